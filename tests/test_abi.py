"""CPU: the C-ABI library loads and exports every symbol include/*.h declares.  No compute
calls (there is no GPU here); the error path for a missing device is exercised instead."""
import ctypes as C
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        names.update(re.findall(r"\b((?:bfhip|convolver)_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_every_declared_symbol_is_exported():
    import brutefir_amd as bf
    lib = C.CDLL(bf.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_struct_layouts_match_the_reference_structs():
    import brutefir_amd as bf
    # struct bfoverflow (bfmod.h:99-104): uint, int32, double, double -> 24 bytes, no padding
    assert C.sizeof(bf.Overflow) == 24
    assert bf.Overflow.largest.offset == 8 and bf.Overflow.max.offset == 16
    assert C.sizeof(bf.Format) == 32 and bf.Format.scale.offset == 16


def test_fails_loudly_without_a_device():
    import brutefir_amd as bf
    if bf.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(bf.BfhipError, match="no HIP device"):
        bf.Engine(1024, 4, 4, 2, 2)


def test_argument_validation_needs_no_device():
    import brutefir_amd as bf
    L = bf.lib()
    assert not L.bfhip_engine_create(0, 1000, 4, 4, 2, 2)       # not a power of two
    assert b"Invalid length" in L.bfhip_last_error()
    assert not L.bfhip_engine_create(0, 1024, 4, 3, 2, 2)       # realsize
    assert b"Invalid real size" in L.bfhip_last_error()


def test_product_never_imports_the_oracle():
    for path in glob.glob(os.path.join(ROOT, "brutefir_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".h", ".hip", ".cpp", ".c")):
            src = open(path, errors="ignore").read()
            assert "bforacle" not in src and "bf_oracle" not in src and "libbfref" not in src, path
