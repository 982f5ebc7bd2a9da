"""Child process of tests/test_abi.py: every bfhip_engine_* / bfhip_nupc_* entry point called with a
NULL handle and NULL / zero arguments.  No entry point may crash; those that return an int
return an error (< 0) or, for the pure queries, 0.  Prints one line per call and SWEEP DONE."""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import brutefir_amd as bf  # noqa: E402

lib = C.CDLL(bf.LIB_PATH)
hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("bfhip.h", "bfhip_nupc.h"))
hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
decls = re.findall(r"\n(int|unsigned int|long|void) (bfhip_(?:engine|nupc)_\w+)\(([^;]*?)\);", hdr, re.S)
for ret, name, args in decls:
    if name in ("bfhip_engine_create", "bfhip_nupc_create"):
        continue
    fn = getattr(lib, name)
    fn.restype = None if ret == "void" else C.c_long
    a = []
    for arg in args.split(","):
        arg = arg.strip()
        if not arg or arg == "void":
            continue
        if "*" in arg or "[" in arg:
            a.append(C.c_void_p(0))
        elif arg.startswith("double"):
            a.append(C.c_double(0.0))
        else:
            a.append(C.c_int(0))
    print(name, end=" ", flush=True)
    r = fn(*a)
    print("void" if ret == "void" else C.c_int(r).value, flush=True)
print("SWEEP DONE")
