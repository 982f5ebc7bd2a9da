"""Build libbfhip.so (the HIP engine + its C ABI) in-tree for gfx950.

    python -m brutefir_amd.build            # rebuild what is older than its sources
    python -m brutefir_amd.build --force

Every translation unit is compiled to an object of its own (in parallel) and linked, so a change
to one file costs one compile.  csrc/host_ops.cpp -- the pure host half of the convolver.h
boundary (what runs in processes that must not own a HIP context: bfconf's parent before the
fork, bflogic_eq) -- is compiled with g++: it cannot contain a HIP call.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libbfhip.so")
INC = os.path.join(HERE, "..", "include")
PUBLIC = [os.path.join("..", "..", "include", h) for h in ("bfhip.h", "bfhip_convolver.h", "bfhip_nupc.h")]
DEVICE = ["kernels.h", "fft_lds.h", "fft_wave.h", "bigfft.h"] + PUBLIC
# translation unit -> what it includes
UNITS = {
    "bfhip.hip": DEVICE + ["conv_shared.h", "alloc.h"],
    "convolver_abi.hip": DEVICE + ["conv_shared.h"],
    "nupc.hip": DEVICE + ["alloc.h"],
    "host_ops.cpp": ["host_fft.h", "conv_shared.h"] + PUBLIC,
}


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _unit_stale(src, obj):
    t = _mtime(obj)
    if t == 0.0:
        return True
    deps = [src] + [d for d in UNITS[src]]
    return any(_mtime(os.path.join(CSRC, d)) > t for d in deps)


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    units = [u for u in UNITS if os.path.exists(os.path.join(CSRC, u))]
    jobs = []
    for u in units:
        obj = os.path.join(OBJ, os.path.splitext(u)[0] + ".o")
        if force or _unit_stale(u, obj):
            if u.endswith(".hip"):
                cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                       os.path.join(CSRC, u), "-o", obj]
            else:
                cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, u), "-o", obj]
            jobs.append(cmd)
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.splitext(u)[0] + ".o") for u in units]
    if force or jobs or not os.path.exists(LIB) or any(_mtime(o) > _mtime(LIB) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
