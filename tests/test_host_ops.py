"""CPU (no GPU anywhere): the pure-host half of the convolver.h boundary (csrc/host_ops.cpp).

BruteFIR prepares coefficients in the PARENT before it forks its filter processes
(bfconf.c:1979-2019) and bflogic_eq renders new ones in a process of its own
(rendereq.h:66-91); neither may initialise HIP.  These entry points therefore run without a
device -- which this container proves, it has none -- and are checked here against the oracle
(itself pinned to the reference's mixnscale/layout code, tests/test_oracle_golden.py) and numpy."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import bforacle as bo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREC = [(4, 3e-6), (8, 1e-13)]


def p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def cv():
    import brutefir_amd as bf
    L = bf.lib()
    L.convolver_coeffs2cbuf.restype = C.c_void_p
    L.convolver_coeffs2cbuf.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    L.convolver_runtime_coeffs2cbuf.argtypes = [C.c_void_p, C.c_void_p]
    L.convolver_fftplan.restype = C.c_void_p
    L.bfhip_fftplan_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.convolver_td_new.restype = C.c_void_p
    L.convolver_td_new.argtypes = [C.c_void_p, C.c_int]
    L.convolver_verify_cbuf.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.convolver_debug_dump_cbuf.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_int]
    L.bfhip_coeff_mark_dirty.argtypes = [C.c_void_p]
    L.bfhip_coeff_mark_dirty.restype = None
    L.bfhip_coeff_dirty_sequence.restype = C.c_ulonglong
    return L


def test_no_device_is_visible_here_or_the_point_is_moot():
    import brutefir_amd as bf
    if bf.device_count() > 0:
        pytest.skip("a GPU is visible; the no-HIP property is shown in the CPU container")
    assert bf.device_count() == 0


def test_host_ops_object_contains_no_hip_call():
    """host_ops.cpp is compiled by g++ (brutefir_amd/build.py); its object must not reference the
    HIP runtime at all"""
    obj = os.path.join(ROOT, "brutefir_amd", "build", "host_ops.o")
    if not os.path.exists(obj):
        pytest.skip("object not kept")
    syms = subprocess.run(["nm", "-u", obj], capture_output=True, text=True).stdout
    assert "hip" not in syms.lower(), [s for s in syms.splitlines() if "hip" in s.lower()]


@pytest.mark.parametrize("rs,tol", PREC)
@pytest.mark.parametrize("L", [4, 64, 1024, 8192])
def test_coeffs2cbuf_without_a_device(cv, rs, tol, L):
    dt = np.float32 if rs == 4 else np.float64
    assert cv.convolver_init(None, L, rs) == 1
    assert cv.convolver_cbufsize() == 2 * L * rs
    o = bo.Ctx(L, rs)
    rng = np.random.default_rng(L + rs)
    for ntaps in (L, L // 2 + 1, 1, 0):
        taps = rng.standard_normal(max(ntaps, 1)).astype(dt)
        dest = np.full(2 * L, 7.0, dt)
        r = cv.convolver_coeffs2cbuf(p(taps), ntaps, 0.5, p(dest))
        assert r == dest.ctypes.data
        want = o.coeffs2cbuf(taps[:ntaps], 0.5) if ntaps else np.zeros(2 * L, dt)
        assert np.abs(dest - want).max() <= tol * 50 * max(np.abs(want).max(), 1e-30) + 1e-300
    # optional_dest == NULL: the library allocates (never freed, like the reference)
    taps = rng.standard_normal(L).astype(dt)
    r = cv.convolver_coeffs2cbuf(p(taps), L, 1.0, None)
    got = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_float if rs == 4 else C.c_double)), (2 * L,))
    want = o.coeffs2cbuf(taps, 1.0)
    assert np.abs(got - want).max() <= tol * 50 * np.abs(want).max()
    # NaN / Inf -> NULL + message (fftw_convolver.c:543-546)
    bad = np.array([1.0, np.inf], dt)
    assert cv.convolver_coeffs2cbuf(p(bad), 2, 1.0, p(np.empty(2 * L, dt))) is None
    # more taps than a partition holds are cut at L (`len`, :533)
    long = rng.standard_normal(L + 9).astype(dt)
    d1, d2 = np.empty(2 * L, dt), np.empty(2 * L, dt)
    cv.convolver_coeffs2cbuf(p(long), L + 9, 1.0, p(d1))
    cv.convolver_coeffs2cbuf(p(long), L, 1.0, p(d2))
    assert np.array_equal(d1, d2)
    # the run-time variant: L reals, no scale
    full = rng.standard_normal(L).astype(dt)
    dest = np.empty(2 * L, dt)
    cv.convolver_runtime_coeffs2cbuf(p(full), p(dest))
    want = o.runtime_coeffs2cbuf(full)
    assert np.abs(dest - want).max() <= tol * 50 * np.abs(want).max()


def test_init_validation_messages(cv, capfd):
    assert cv.convolver_init(b"wisdom-ignored", 1000, 4) == 0
    assert cv.convolver_init(b"wisdom-ignored", 1024, 3) == 0
    assert cv.convolver_init(b"wisdom-ignored", 2, 4) == 0
    err = capfd.readouterr().err
    assert "Invalid length 1000." in err and "Invalid real size 3." in err
    assert cv.convolver_init(b"wisdom-ignored", 1024, 8) == 1


@pytest.mark.parametrize("rs,tol", PREC)
def test_fftplan_execute_is_fftw_r2hc_hc2r_by_definition(cv, rs, tol):
    dt = np.float32 if rs == 4 else np.float64
    assert cv.convolver_init(None, 256, rs) == 1
    rng = np.random.default_rng(rs)
    for order in (1, 2, 3, 5, 9, 14):
        n = 1 << order
        x = rng.standard_normal(n).astype(dt)
        hc = np.empty(n, dt)
        cv.bfhip_fftplan_execute(cv.convolver_fftplan(order, 0, 0), p(x), p(hc))
        X = np.fft.rfft(x.astype(np.float64))
        got = np.zeros(n // 2 + 1, complex)
        got.real = hc[:n // 2 + 1]
        got.imag[1:n // 2] = hc[n - 1:n // 2:-1]
        assert np.abs(got - X).max() <= tol * 20 * max(np.abs(X).max(), 1)
        back = hc.copy()
        cv.bfhip_fftplan_execute(cv.convolver_fftplan(order, 1, 1), p(back), p(back))   # in place
        assert np.abs(back / n - x).max() <= tol * 50
    assert cv.convolver_fftplan(5, 0, 0) == cv.convolver_fftplan(5, 0, 1)              # cached, never freed


@pytest.mark.parametrize("rs,tol", PREC)
def test_debug_dump_is_the_inverse_of_coeffs2cbuf(cv, rs, tol, tmp_path):
    """convolver_debug_dump_cbuf (fftw_convolver.c:624-660) writes the taps back as text:
    cbuf -> mixnscale(OUTPUT) -> HC2R -> second half, "%.16e" per line"""
    dt = np.float32 if rs == 4 else np.float64
    L = 256
    assert cv.convolver_init(None, L, rs) == 1
    rng = np.random.default_rng(77 + rs)
    taps = [rng.standard_normal(L).astype(dt) for _ in range(3)]
    cbufs = [np.empty(2 * L, dt) for _ in taps]
    for t, c in zip(taps, cbufs):
        assert cv.convolver_coeffs2cbuf(p(t), L, 1.0, p(c)) == c.ctypes.data
    arr = (C.c_void_p * 3)(*[c.ctypes.data for c in cbufs])
    assert cv.convolver_verify_cbuf(arr, 3) == 1
    path = tmp_path / "dump.txt"
    cv.convolver_debug_dump_cbuf(str(path).encode(), arr, 3)
    lines = path.read_text().split()
    assert len(lines) == 3 * L
    got = np.array([float(v) for v in lines]).reshape(3, L)
    for k in range(3):
        assert np.abs(got[k] - taps[k]).max() <= tol * 100
    # verify_cbuf: one NaN anywhere fails the lot, with the reference's message
    cbufs[1][17] = np.nan
    assert cv.convolver_verify_cbuf(arr, 3) == 0
    assert cv.convolver_verify_cbuf(arr, 1) == 1


def test_change_notices_cross_a_fork(cv):
    """bfhip_coeff_mark_dirty in a fork()ed child (bflogic_eq's position) is seen by the parent
    (the filter process's position) through the MAP_SHARED table convolver_init() created"""
    import mmap
    L = 64
    assert cv.convolver_init(None, L, 4) == 1
    shm = mmap.mmap(-1, 2 * L * 4)                      # shared, like bfconf's coefficient segment
    dest = np.frombuffer(shm, np.float32)
    before = cv.bfhip_coeff_dirty_sequence()
    taps = np.arange(L, dtype=np.float32)
    pid = os.fork()
    if pid == 0:
        try:
            cv.convolver_runtime_coeffs2cbuf(p(taps), C.c_void_p(dest.ctypes.data))
        finally:
            os._exit(0)
    assert os.waitpid(pid, 0)[1] == 0
    assert cv.bfhip_coeff_dirty_sequence() == before + 1
    want = bo.Ctx(L, 4).runtime_coeffs2cbuf(taps)
    assert np.abs(dest - want).max() <= 1e-5 * np.abs(want).max()
    # a direct writer announces its change itself
    cv.bfhip_coeff_mark_dirty(C.c_void_p(dest.ctypes.data))
    assert cv.bfhip_coeff_dirty_sequence() == before + 2


def test_td_new_needs_no_device(cv):
    assert cv.convolver_init(None, 256, 4) == 1
    taps = np.ones(5, np.float32)
    assert cv.convolver_td_new(p(taps), 5)
    assert cv.convolver_td_new(p(taps), 0) is None


def test_host_ops_under_address_and_ub_sanitizers(tmp_path):
    """The GPU pool has no device sanitizer, the host half can have one: csrc/host_ops.cpp and a
    driver that walks its entry points (tests/chost/host_ops_driver.cpp) are compiled with
    -fsanitize=address,undefined and run; any finding aborts the run."""
    exe = str(tmp_path / "host_ops_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "brutefir_amd", "csrc", "host_ops.cpp"),
           os.path.join(ROOT, "tests", "chost", "host_ops_driver.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("no sanitizer runtime in this toolchain: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))     # coefficient memory is never freed by design
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "host_ops_driver: clean" in r.stdout
