"""CPU: the C-ABI library loads and exports every symbol include/*.h declares.  No compute
calls (there is no GPU here); the error path for a missing device is exercised instead."""
import ctypes as C
import glob
import os
import sys
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


def test_host_delay_machine_matches_delay_c_without_a_device():
    """the engine's host mirror of delay.c's buffer machine (it emits the moves the device then
    executes) against the oracle's delay, which is bit-exact against the reference's own delay.c
    (tests/test_oracle_delay.py): random fragment sizes, sample sizes, delay histories across the
    short / long regimes, unlimited maxdelay"""
    import ctypes as C
    import numpy as np
    import brutefir_amd as bf
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bforacle as bo
    L = bf.lib()
    L.bfhip_selftest_delay_new.restype = C.c_void_p
    L.bfhip_selftest_delay_new.argtypes = [C.c_int] * 4
    L.bfhip_selftest_delay_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bfhip_selftest_delay_free.argtypes = [C.c_void_p]
    O = bo.lib()
    O.bfo_delay_new.restype = C.c_void_p
    O.bfo_delay_new.argtypes = [C.c_int] * 4
    O.bfo_delay_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    O.bfo_delay_free.argtypes = [C.c_void_p]
    for seed in range(60):
        rng = np.random.default_rng(seed)
        F = int(rng.choice([16, 64, 128, 256]))
        ss = int(rng.choice([1, 2, 4, 8]))
        maxd = int(rng.choice([0, F // 2, F, 3 * F + 5, 10 * F, -1]))
        lim = 12 * F if maxd < 0 else maxd
        init = int(rng.integers(0, lim + 1))
        d = L.bfhip_selftest_delay_new(F, init, maxd, ss)
        o = O.bfo_delay_new(F, init, maxd, ss)
        assert d and o
        delay = init
        for b in range(40):
            if rng.random() < 0.3:
                delay = int(rng.integers(0, lim + 1))
            x = rng.integers(0, 256, F * ss, dtype=np.uint8)
            a, c = x.copy(), x.copy()
            assert L.bfhip_selftest_delay_update(d, a.ctypes.data, delay) >= 0, L.bfhip_last_error()
            O.bfo_delay_update(o, c.ctypes.data, delay)
            assert np.array_equal(a, c), (seed, b, F, ss, maxd, init, delay)
        L.bfhip_selftest_delay_free(d)
        O.bfo_delay_free(o)


def test_no_entry_point_crashes_on_a_null_handle():
    """tests/helpers/null_sweep.py in a child process (a crash must not take pytest with it): all
    bfhip_engine_* / bfhip_nupc_* calls with a NULL handle return an error or a neutral 0"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "null_sweep.py")],
                       capture_output=True, text=True, timeout=300)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines and lines[-1] == "SWEEP DONE", (r.returncode, lines[-3:], r.stderr[-1000:])
    calls = dict(ln.split() for ln in lines[:-1])
    assert len(calls) >= 70
    neutral = {"bfhip_engine_output_lag", "bfhip_engine_blockcounter", "bfhip_engine_uses_wave_fft",
               "bfhip_engine_uses_stream_layout", "bfhip_engine_ring_depth", "bfhip_nupc_taps", "bfhip_nupc_latency",
               "bfhip_engine_coeff_is_resident", "bfhip_engine_output_is_active", "bfhip_engine_uses_diag_mac", "bfhip_engine_pair_launches"}
    for name, ret in calls.items():
        if ret == "void":
            continue
        assert (int(ret) == 0) if name in neutral else (int(ret) < 0), (name, ret)
