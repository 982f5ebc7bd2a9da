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


def test_nonuniform_schedule_validation_needs_no_device():
    """bfhip_nupc_create rejects a schedule whose long partitions could not be ready in time
    before it touches the device (include/bfhip_nupc.h)"""
    import brutefir_amd as bf
    with pytest.raises(bf.BfhipError, match="not be ready in time"):
        bf.Nupc([64, 1024], [2, 4], 4, 1, 1)
    with pytest.raises(bf.BfhipError, match="ascend"):
        bf.Nupc([128, 64], [2, 2], 4, 1, 1)
    if bf.device_count() == 0:
        with pytest.raises(bf.BfhipError, match="no HIP device"):
            bf.Nupc([64, 128], [2, 2], 4, 1, 1)


def test_realtime_entry_points_need_a_finalized_engine():
    import brutefir_amd as bf
    L = bf.lib()
    assert L.bfhip_engine_rt_submit(None, None) < 0
    assert b"not in real-time mode" in L.bfhip_last_error()
    assert L.bfhip_engine_rt_buffer(None, 0, 0) is None


def test_product_never_imports_the_oracle():
    for path in glob.glob(os.path.join(ROOT, "brutefir_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".h", ".hip", ".cpp", ".c")):
            src = open(path, errors="ignore").read()
            assert "bforacle" not in src and "bf_oracle" not in src and "libbfref" not in src, path
