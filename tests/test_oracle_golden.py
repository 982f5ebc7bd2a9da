"""CPU: the oracle replayed against golden vectors produced by the REFERENCE'S OWN inner loops
(tests/golden/make_golden.py -> oracle/_ref).  Everything here is bit-exact: these loops have
no FFT in them, and the oracle is compiled without FMA contraction like the reference."""
import ctypes as C
import os

import numpy as np
import pytest

import bforacle as bo

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PREC = [(4, "f32"), (8, "f64")]


def load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("rs,tag", PREC)
def test_mix_and_products_bit_exact(rs, tag):
    g = load("ref_ops_%s.npz" % tag)
    c = bo.Ctx(int(g["L"]), rs)
    for n in (1, 2, 3, 4, 6):
        bufs, sc = g["mix%d_in" % n], g["mix%d_scales" % n]
        assert np.array_equal(c.mixnscale(list(bufs), sc, bo.MIX_INPUT), g["mix%d_input" % n])
        assert np.array_equal(c.mixnscale(list(bufs), sc, bo.MIX_OUTPUT), g["mix%d_output" % n])
    b, h, d = g["b"], g["h"], g["d"]
    assert np.array_equal(c.convolve(b, h), g["convolve"])
    assert np.array_equal(c.convolve_inplace(b, h), g["convolve_inplace"])
    assert np.array_equal(c.convolve_add(b, h, d), g["convolve_add"])
    # the SSE/SSE2 kernels of convolver_xmm.c compute the same bits as the C loop
    assert np.array_equal(g["convolve_add_simd"], g["convolve_add"])
    assert np.array_equal(c.dirac_convolve(b), g["dirac_convolve"])


FORMATS = ["S8", "S16_LE", "S16_BE", "S24_LE", "S24_BE", "S24_4LE", "S24_4BE", "S32_LE",
           "S32_BE", "FLOAT_LE", "FLOAT_BE", "FLOAT64_LE", "FLOAT64_BE"]


@pytest.mark.parametrize("rs,tag", PREC)
@pytest.mark.parametrize("name", FORMATS)
def test_sample_conversion_bit_exact(rs, tag, name):
    g = load("ref_conv_%s.npz" % tag)
    L, spacing = 64, 3
    nbytes, sbytes, isfloat, le = bo.SAMPLE_FORMATS[name]
    c = bo.Ctx(L, rs)
    real = c.raw2real(g["r2r_%s_raw" % name], nbytes, isfloat, spacing, 0 if le else 1, L)
    assert np.array_equal(real, g["r2r_%s_real" % name])
    fmt = bo.make_format(name, spacing, 0)
    full = float(1 << (8 * sbytes - 1)) if not isfloat else 1.0
    of = bo.Overflow(0, 0, 0.0, 1.0 if isfloat else full - 1)
    st, raw = c.cbuf2raw(g["rr_%s_x" % name], fmt, L * spacing * nbytes, of)
    assert st == 0
    assert np.array_equal(raw, g["rr_%s_raw" % name])
    assert of.astuple() == tuple(g["rr_%s_of" % name])


@pytest.mark.parametrize("rs,tag", PREC)
def test_dither_bit_exact(rs, tag):
    g = load("ref_dither_%s.npz" % tag)
    L = 64
    c = bo.Ctx(L, rs)
    n_ch = g["x"].shape[1]
    assert c.dither_init(n_ch, int(g["rate"]))
    tab = c.dither_table()
    assert len(tab) == int(g["table_size"])
    assert np.array_equal(tab[:4096], g["table_head"])
    fmt = bo.make_format("S16_LE", 1, 0)
    of = [bo.Overflow(0, 0, 0.0, 32767.0) for _ in range(n_ch)]
    for b in range(g["x"].shape[0]):
        for ch in range(n_ch):
            st, raw = c.cbuf2raw(g["x"][b, ch], fmt, L * 2, of[ch], dither_channel=ch)
            assert st == 0
            assert np.array_equal(raw.view(np.int16), g["raw"][b, ch]), (b, ch)
            assert bo.lib().bfo_dither_randtab_ptr(c.h, ch) == g["randtab_ptr"][b, ch]
    for ch in range(n_ch):
        assert of[ch].astuple() == tuple(g["of"][ch])


def test_live_reference_agrees_with_goldens():
    """where oracle/_ref is built (this container), the live reference code reproduces the
    committed fixtures -- guards against a stale fixture"""
    R = bo.ref()
    if R is None:
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    g = load("ref_ops_f32.npz")
    R.ref_set_length(int(g["L"]), 0.0)
    b, h, d = g["b"].copy(), g["h"].copy(), g["d"].copy()
    R.ref_convolve_add(4, b.ctypes.data, h.ctypes.data, d.ctypes.data)
    assert np.array_equal(d, g["convolve_add"])
