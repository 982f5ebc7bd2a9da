"""GPU: the 22 convolver.h link-time symbols exported by libbfhip.so (include/bfhip_convolver.h),
called exactly as the C host would (host buffers, reference layouts).
 * FFT-free ops: BIT-EXACT against the golden vectors produced by the reference's own code
   (tests/golden/ref_*.npz) -- same bits as fftw_convfuns.h / raw2real.h / real2raw.h.
 * FFT ops: against the oracle / numpy within 1e-5 (f32) / 1e-12 (f64) relative."""
import ctypes as C
import os

import numpy as np
import pytest

import bforacle as bo

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PREC = [(4, "f32", 3e-6), (8, "f64", 1e-13)]


class SampleFormat(C.Structure):
    _fields_ = [("isfloat", C.c_int), ("swap", C.c_int), ("bytes", C.c_int), ("sbytes", C.c_int),
                ("scale", C.c_double), ("format", C.c_int)]


class BufferFormat(C.Structure):
    """struct buffer_format, dai.h:30-34"""
    _fields_ = [("sf", SampleFormat), ("sample_spacing", C.c_int), ("byte_offset", C.c_int)]


def p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture()
def cv(hip):
    L = hip.lib()
    L.convolver_coeffs2cbuf.restype = C.c_void_p
    L.convolver_coeffs2cbuf.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    L.convolver_mixnscale.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]
    L.convolver_fftplan.restype = C.c_void_p
    L.convolver_td_new.restype = C.c_void_p
    L.convolver_td_new.argtypes = [C.c_void_p, C.c_int]
    L.convolver_td_convolve.argtypes = [C.c_void_p, C.c_void_p]
    L.bfhip_fftplan_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.convolver_raw2cbuf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(BufferFormat), C.c_void_p, C.c_void_p]
    L.convolver_cbuf2raw.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BufferFormat), C.c_int, C.c_void_p, C.c_void_p]
    for f in ("convolver_time2freq", "convolver_freq2time", "convolver_convolve_inplace",
              "convolver_dirac_convolve", "convolver_runtime_coeffs2cbuf"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
    for f in ("convolver_convolve", "convolver_convolve_add", "convolver_crossfade_inplace",
              "convolver_convolve_eval"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.convolver_dirac_convolve_inplace.argtypes = [C.c_void_p]
    return L


def test_init_validation_and_cbufsize(cv):
    assert cv.convolver_init(b"ignored-wisdom", 1000, 4) == 0
    assert cv.convolver_init(b"ignored-wisdom", 1024, 3) == 0
    assert cv.convolver_init(b"ignored-wisdom", 1024, 8) == 1
    assert cv.convolver_cbufsize() == 2 * 1024 * 8
    assert cv.convolver_td_block_length(0) == -1
    assert [cv.convolver_td_block_length(n) for n in (1, 2, 3, 31, 32, 33)] == [1, 2, 4, 32, 32, 64]


@pytest.mark.parametrize("rs,tag,_", PREC)
def test_mix_and_products_bit_exact_vs_reference(cv, rs, tag, _):
    g = np.load(os.path.join(G, "ref_ops_%s.npz" % tag))
    L = int(g["L"])
    dt = np.float32 if rs == 4 else np.float64
    assert cv.convolver_init(None, L, rs) == 1
    for n in (1, 2, 3, 4, 6):
        bufs = np.ascontiguousarray(g["mix%d_in" % n])
        sc = g["mix%d_scales" % n]
        arr = (C.c_void_p * n)(*[bufs[i].ctypes.data for i in range(n)])
        for mode, key in ((1, "input"), (3, "output")):
            out = np.empty(2 * L, dt)
            cv.convolver_mixnscale(arr, p(out), (C.c_double * n)(*sc), n, mode)
            assert np.array_equal(out, g["mix%d_%s" % (n, key)]), (n, key)
    b, h, d = g["b"].copy(), g["h"].copy(), g["d"].copy()
    out = np.empty(2 * L, dt)
    cv.convolver_convolve(p(b), p(h), p(out))
    assert np.array_equal(out, g["convolve"])
    out = b.copy()
    cv.convolver_convolve_inplace(p(out), p(h))
    assert np.array_equal(out, g["convolve_inplace"])
    out = d.copy()
    cv.convolver_convolve_add(p(b), p(h), p(out))
    assert np.array_equal(out, g["convolve_add"])
    out = np.empty(2 * L, dt)
    cv.convolver_dirac_convolve(p(b), p(out))
    assert np.array_equal(out, g["dirac_convolve"])
    out = b.copy()
    cv.convolver_dirac_convolve_inplace(p(out))
    assert np.array_equal(out, g["dirac_convolve"])


FORMATS = ["S8", "S16_LE", "S16_BE", "S24_LE", "S24_BE", "S24_4LE", "S24_4BE", "S32_LE",
           "S32_BE", "FLOAT_LE", "FLOAT_BE", "FLOAT64_LE", "FLOAT64_BE"]


@pytest.mark.parametrize("rs,tag,_", PREC)
def test_sample_conversion_bit_exact_vs_reference(cv, rs, tag, _):
    g = np.load(os.path.join(G, "ref_conv_%s.npz" % tag))
    L, spacing = 64, 3
    dt = np.float32 if rs == 4 else np.float64
    assert cv.convolver_init(None, L, rs) == 1
    for name in FORMATS:
        nbytes, sbytes, isfloat, le = bo.SAMPLE_FORMATS[name]
        bf = BufferFormat(SampleFormat(isfloat, 0 if le else 1, nbytes, sbytes, 1.0, 0), spacing, 0)
        cbuf, nxt = np.zeros(2 * L, dt), np.zeros(2 * L, dt)
        raw = g["r2r_%s_raw" % name].copy()
        cv.convolver_raw2cbuf(p(raw), p(cbuf), p(nxt), C.byref(bf), None, None)
        assert np.array_equal(nxt[:L], g["r2r_%s_real" % name]), name
        assert np.array_equal(cbuf[L:], g["r2r_%s_real" % name]), name      # the sliding window copy
        full = float(1 << (8 * sbytes - 1)) if not isfloat else 1.0
        of = bo.Overflow(0, 0, 0.0, 1.0 if isfloat else full - 1)
        x = np.zeros(2 * L, dt)
        x[:L] = g["rr_%s_x" % name]
        out = np.zeros(L * spacing * nbytes, np.uint8)
        cv.convolver_cbuf2raw(p(x), p(out), C.byref(bf), 0, None, C.byref(of))
        assert np.array_equal(out, g["rr_%s_raw" % name]), name
        assert of.astuple() == tuple(g["rr_%s_of" % name]), name


@pytest.mark.parametrize("rs,tag,tol", PREC)
@pytest.mark.parametrize("L", [64, 1024, 8192, 16384, 65536, 262144])
def test_fft_ops_vs_definition(cv, rs, tag, tol, L):
    dt = np.float32 if rs == 4 else np.float64
    assert cv.convolver_init(None, L, rs) == 1
    rng = np.random.default_rng(L + rs)
    x = rng.standard_normal(2 * L).astype(dt)
    hc = np.empty(2 * L, dt)
    cv.convolver_time2freq(p(x), p(hc))
    X = np.fft.rfft(x.astype(np.float64))
    assert np.abs(bo.hc_to_complex(hc) - X).max() <= tol * np.abs(X).max() * np.log2(2 * L)
    back = hc.copy()
    cv.convolver_freq2time(p(back), p(back))                      # in place, like the reference
    assert np.abs(back / (2 * L) - x).max() <= tol * 10 * np.log2(2 * L)
    # coefficient preparation against the oracle (layout, 1/n_fft, scale, zero padding)
    o = bo.Ctx(L, rs)
    taps = rng.standard_normal(L // 2 + 3).astype(dt)
    dest = np.empty(2 * L, dt)
    r = cv.convolver_coeffs2cbuf(p(taps), len(taps), 0.5, p(dest))
    assert r == dest.ctypes.data
    want = o.coeffs2cbuf(taps, 0.5)
    assert np.abs(dest - want).max() <= tol * 50 * np.abs(want).max()
    full = rng.standard_normal(L).astype(dt)
    cv.convolver_runtime_coeffs2cbuf(p(full), p(dest))
    want = o.runtime_coeffs2cbuf(full)
    assert np.abs(dest - want).max() <= tol * 50 * np.abs(want).max()
    bad = np.array([1.0, np.nan], dt)
    assert cv.convolver_coeffs2cbuf(p(bad), 2, 1.0, p(dest)) is None
    # crossfade and cascade evaluation against the oracle
    qa = o.coeffs2cbuf(rng.standard_normal(L).astype(dt))
    qb = o.coeffs2cbuf(rng.standard_normal(L).astype(dt))
    want = o.crossfade_inplace(qa, qb)
    a2, b2, buf = qa.copy(), qb.copy(), np.zeros(2 * L, dt)
    cv.convolver_crossfade_inplace(p(a2), p(b2), p(buf))
    assert np.abs(a2 - want).max() <= tol * 100 * np.abs(want).max()
    st_o, st_g = np.zeros(3 * L, dt), np.zeros(3 * L, dt)
    for _ in range(3):
        hcx = o.time2freq(rng.standard_normal(2 * L).astype(dt)) / dt(2 * L)
        want = o.convolve_eval(hcx.astype(dt), st_o)
        got = np.empty(2 * L, dt)
        cv.convolver_convolve_eval(p(hcx.astype(dt)), p(st_g), p(got))
        assert np.abs(got - want).max() <= tol * 100 * np.abs(want).max()
        assert np.abs(st_g[:L] - st_o[:L]).max() <= tol * 100


@pytest.mark.parametrize("rs,tag,tol", PREC)
def test_fftplan_and_td_convolve(cv, rs, tag, tol):
    """bflogic_eq's plan handle and delay.c's small overlap-save (convolver_td_*)"""
    dt = np.float32 if rs == 4 else np.float64
    assert cv.convolver_init(None, 256, rs) == 1
    rng = np.random.default_rng(rs)
    for order in (1, 2, 5, 9):
        n = 1 << order
        x = rng.standard_normal(n).astype(dt)
        hc = np.empty(n, dt)
        cv.bfhip_fftplan_execute(cv.convolver_fftplan(order, 0, 0), p(x), p(hc))
        X = np.fft.rfft(x.astype(np.float64))
        got = np.zeros(n // 2 + 1, complex)
        got.real = hc[:n // 2 + 1]
        got.imag[1:n // 2] = hc[n - 1:n // 2:-1]
        assert np.abs(got - X).max() <= tol * 20 * max(np.abs(X).max(), 1)
        back = np.empty(n, dt)
        cv.bfhip_fftplan_execute(cv.convolver_fftplan(order, 1, 1), p(hc), p(back))
        assert np.abs(back / n - x).max() <= tol * 50
    for ntaps in (1, 5, 31, 64):
        taps = rng.standard_normal(ntaps).astype(dt)
        tdc = cv.convolver_td_new(p(taps), ntaps)
        blk = cv.convolver_td_block_length(ntaps)
        sig = rng.standard_normal(2 * blk).astype(dt)
        buf = sig.copy()
        cv.convolver_td_convolve(tdc, p(buf))
        # overlap-save with the taps in the 2nd half of the window: first `blk` samples valid
        want = np.convolve(sig.astype(np.float64), taps.astype(np.float64))[blk:2 * blk]
        assert np.abs(buf[:blk] - want).max() <= tol * 200 * max(np.abs(want).max(), 1)
