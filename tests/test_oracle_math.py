"""CPU: the oracle's FFT-dependent ops against their mathematical definition (numpy.fft,
numpy.convolve) and analytic known-answer tests.  FFTW itself is absent from this image
(SURVEY 8c), so these ops are pinned by definition, not by reference output."""
import numpy as np
import pytest

import bforacle as bo

PREC = [(4, 3e-6, 2e-6), (8, 1e-14, 1e-14)]


@pytest.mark.parametrize("rs,tol,_", PREC)
@pytest.mark.parametrize("L", [4, 64, 1024, 8192])
def test_r2hc_hc2r_definition(rs, tol, _, L):
    rng = np.random.default_rng(L)
    c = bo.Ctx(L, rs)
    x = rng.standard_normal(2 * L).astype(c.dt)
    hc = c.time2freq(x)
    X = np.fft.rfft(x.astype(np.float64))
    assert np.abs(bo.hc_to_complex(hc) - X).max() <= tol * np.abs(X).max() * np.log2(2 * L)
    # HC2R is the unnormalised inverse: hc2r(r2hc(x)) = n_fft * x
    back = c.freq2time(hc)
    assert np.abs(back / (2 * L) - x).max() <= tol * 10 * np.log2(2 * L)


@pytest.mark.parametrize("rs,tol,_", PREC)
def test_coeffs2cbuf_layout_and_scaling(rs, tol, _):
    """partition = spectrum of [L zeros | taps*scale] / n_fft in the reordered layout"""
    L = 256
    rng = np.random.default_rng(3)
    c = bo.Ctx(L, rs)
    taps = rng.standard_normal(100).astype(c.dt)
    q = c.coeffs2cbuf(taps, scale=0.5)
    pad = np.zeros(2 * L)
    pad[L:L + 100] = taps.astype(np.float64) * 0.5
    X = np.fft.rfft(pad) / (2 * L)
    assert np.abs(bo.reordered_to_complex(q) - X).max() <= tol * 50 * np.abs(X).max()
    assert q[4] == pytest.approx(X[L].real, abs=tol)       # Nyquist in the Im[0] slot
    # runtime variant: L taps, no scale
    full = rng.standard_normal(L).astype(c.dt)
    assert np.allclose(c.runtime_coeffs2cbuf(full), c.coeffs2cbuf(full, 1.0), rtol=0, atol=0)
    assert c.coeffs2cbuf(np.array([1.0, np.nan])) is None


def test_coeffs2cbuf_rejects_bad_init():
    with pytest.raises(ValueError):
        bo.Ctx(100, 4)          # not a power of two (fftw_convolver.c:800-803)
    with pytest.raises(ValueError):
        bo.Ctx(64, 2)           # realsize must be 4 or 8 (:796-799)


@pytest.mark.parametrize("rs,tol,_", PREC)
def test_one_partition_is_linear_convolution(rs, tol, _):
    """time2freq -> mixnscale(INPUT) -> convolve -> mixnscale(OUTPUT) -> freq2time: first L
    samples = linear convolution of [prev|cur] with the taps, evaluated for cur"""
    L = 128
    rng = np.random.default_rng(9)
    c = bo.Ctx(L, rs)
    x = rng.standard_normal(2 * L).astype(c.dt)
    h = rng.standard_normal(L).astype(c.dt)
    ring = c.mixnscale([c.time2freq(x)], [1.0], bo.MIX_INPUT)
    y = c.freq2time(c.mixnscale([c.convolve(ring, c.coeffs2cbuf(h))], [1.0], bo.MIX_OUTPUT))
    want = np.convolve(x.astype(np.float64), h.astype(np.float64))[L:2 * L]
    assert np.abs(y[:L] - want).max() <= tol * 200 * np.abs(want).max()


@pytest.mark.parametrize("rs,tol,_", PREC)
def test_dirac_is_identity(rs, tol, _):
    L = 64
    rng = np.random.default_rng(2)
    c = bo.Ctx(L, rs)
    x = rng.standard_normal(2 * L).astype(c.dt)
    ring = c.mixnscale([c.time2freq(x)], [1.0], bo.MIX_INPUT)
    y = c.freq2time(c.mixnscale([c.dirac_convolve(ring)], [1.0], bo.MIX_OUTPUT))
    assert np.abs(y[:L] - x[L:]).max() <= tol * 20
    # and equals convolution with an explicit unit pulse coefficient set
    unit = np.zeros(L, c.dt)
    unit[0] = 1
    y2 = c.freq2time(c.mixnscale([c.convolve(ring, c.coeffs2cbuf(unit))], [1.0], bo.MIX_OUTPUT))
    assert np.abs(y2[:L] - y[:L]).max() <= tol * 20


@pytest.mark.parametrize("rs,tol,_", PREC)
def test_convolve_eval_rewindows(rs, tol, _):
    """cascade re-evaluation: output spectrum k = rfft([valid(y_{k-1}) | valid(y_k)])"""
    L = 64
    rng = np.random.default_rng(4)
    c = bo.Ctx(L, rs)
    state = np.zeros(3 * L, c.dt)
    prev = np.zeros(L)
    for _k in range(3):
        t = rng.standard_normal(2 * L).astype(c.dt)          # a time block, first half valid
        hc = c.time2freq(t) / c.dt(2 * L)                    # so that hc2r returns t
        out = c.convolve_eval(hc.astype(c.dt), state)
        want = np.fft.rfft(np.concatenate([prev, t[:L].astype(np.float64)]))
        assert np.abs(bo.hc_to_complex(out) - want).max() <= tol * 100 * np.abs(want).max()
        prev = t[:L].astype(np.float64)


@pytest.mark.parametrize("rs,tol,_", PREC)
def test_crossfade_is_linear_ramp(rs, tol, _):
    """first L samples: old*(1-n/(L-1)) + new*n/(L-1); samples L..2L-1 stay the new result's"""
    L = 64
    rng = np.random.default_rng(6)
    c = bo.Ctx(L, rs)
    told = rng.standard_normal(2 * L).astype(c.dt)
    tnew = rng.standard_normal(2 * L).astype(c.dt)
    inv = 1.0 / (2 * L)
    qold = c.mixnscale([c.time2freq(told)], [inv], bo.MIX_INPUT)
    qnew = c.mixnscale([c.time2freq(tnew)], [inv], bo.MIX_INPUT)
    q = c.crossfade_inplace(qnew, qold)
    y = c.freq2time(c.mixnscale([q], [1.0], bo.MIX_OUTPUT))
    w = np.arange(L) / (L - 1.0)
    want = np.concatenate([told[:L] * (1 - w) + tnew[:L] * w, tnew[L:]])
    assert np.abs(y - want).max() <= tol * 100
