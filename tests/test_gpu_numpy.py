"""GPU: the HIP engine against an ORACLE-INDEPENDENT model -- float64 linear convolution done by
scipy/numpy (pocketfft), nothing from oracle/ involved.  The oracle restates the reference's
block machinery; these tests pin the same device results a second, independent way, so the rows
whose oracle side cannot be pinned by reference output (cross-fade A7, cascade A8, ring / delay
bookkeeping A12, coefficient preparation A13: FFTW is absent) do not hang on one
restatement only.  Tolerances: north_star's 1e-5 relative RMS for float32, 1e-11 for float64."""
import os

import numpy as np
import pytest
from scipy.signal import fftconvolve

import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _conv(x, h, n):
    return fftconvolve(np.asarray(x, np.float64), np.asarray(h, np.float64))[:n]


def _run_float(e, x, L, O, rs):
    dt = np.float32 if rs == 4 else np.float64
    outs = []
    for b in range(len(x) // L):
        st, raw = e.block(x[b * L:(b + 1) * L])
        assert st == 0, b
        outs.append(np.frombuffer(raw.tobytes(), dt).reshape(L, O))
    return np.concatenate(outs).astype(np.float64)


def test_config_b_full_size_is_linear_convolution(hip):
    """BASELINE.json configs[1]: 8-in/8-out crossbar, 65536 taps = 8192 x 8, float32"""
    L, N, I, O = 8192, 8, 8, 8
    e, irs = cases.crossbar(hip.Engine, L, N, 4, I, O, "S24_4LE", "FLOAT_LE")
    nblk = 2 * N + 3
    x = np.concatenate(cases.raw_blocks(5, nblk, L, I, "S24_4LE"))
    y = _run_float(e, x, L, O, 4)
    xs = x.astype(np.float64) / 8388608.0
    want = np.zeros_like(y)
    for (o, i), h in irs.items():
        want[:, o] += _conv(xs[:, i], h.astype(np.float32), len(x))
    for o in range(O):
        assert cases.rel_rms(y[:, o], want[:, o]) <= 1e-5, o


@pytest.mark.parametrize("rs", [4, 8])
def test_configs0_literal_2x2_4096_taps_single_partition(hip, rs):
    """BASELINE.json configs[0] as worded: 2-in/2-out, 4096 taps, ONE partition"""
    L, N, I, O = 4096, 1, 2, 2
    ofmt = "FLOAT_LE" if rs == 4 else "FLOAT64_LE"
    e, irs = cases.crossbar(hip.Engine, L, N, rs, I, O, "S24_4LE", ofmt)
    x = np.concatenate(cases.raw_blocks(11, 7, L, I, "S24_4LE"))
    y = _run_float(e, x, L, O, rs)
    dt = np.float32 if rs == 4 else np.float64
    xs = x.astype(np.float64) / 8388608.0
    want = np.zeros_like(y)
    for (o, i), h in irs.items():
        want[:, o] += _conv(xs[:, i], h.astype(dt), len(x))
    assert cases.rel_rms(y, want) <= (1e-5 if rs == 4 else 1e-11)


@pytest.mark.parametrize("rs", [4, 8])
def test_delay_cascade_and_crossfade_network_vs_numpy(hip, rs):
    """delays, a cascade with mixed channel + filter inputs and a cross-fading coefficient switch
    in one network (cases.fade_cascade_network) against float64 numpy -- no oracle involved"""
    y, want, L, sw = cases.fade_cascade_network(hip.Engine, rs)
    tol = 1e-5 if rs == 4 else 1e-11
    for ch in range(2):
        assert cases.rel_rms(y[:, ch], want[:, ch]) <= tol, ch
    for b in (sw - 1, sw, sw + 1):                         # the switch block and its neighbours
        s = slice(b * L, (b + 1) * L)
        assert cases.rel_rms(y[s, 0], want[s, 0]) <= 2 * tol, b


def test_xtc_config_with_the_reference_s_own_taps(hip):
    """xtc_config as shipped: `filter_length: 64,64`, the 4096-tap directpath.txt / crosspath.txt
    the reference distributes (tests/golden/xtc_taps.npz holds their numbers), a symmetric 2x2
    cross-talk canceller.  Float outputs against float64 convolution of the very same taps."""
    t = np.load(os.path.join(G, "xtc_taps.npz"))
    direct, cross = t["directpath"], t["crosspath"]
    assert direct.shape == cross.shape == (4096,)
    L, N = 64, 64
    e = hip.Engine(L, N, 4, 2, 2)
    e.set_interleaved(0, "S24_4LE")
    e.set_interleaved(1, "FLOAT_LE")
    cd, cx = e.add_coeff(direct.astype(np.float32)), e.add_coeff(cross.astype(np.float32))
    e.add_filter(in_ch=[0], out_ch=[0], coeff=cd)          # left speaker direct path
    e.add_filter(in_ch=[1], out_ch=[0], coeff=cx)          # left speaker cross path
    e.add_filter(in_ch=[1], out_ch=[1], coeff=cd)          # right speaker direct path
    e.add_filter(in_ch=[0], out_ch=[1], coeff=cx)          # right speaker cross path
    e.finalize()
    nblk = 2 * N + 9
    x = np.concatenate(cases.raw_blocks(17, nblk, L, 2, "S24_4LE", amplitude=0.05))
    y = _run_float(e, x, L, 2, 4)
    xs = x.astype(np.float64) / 8388608.0
    d32, c32 = direct.astype(np.float32), cross.astype(np.float32)
    want = np.stack([_conv(xs[:, 0], d32, len(x)) + _conv(xs[:, 1], c32, len(x)),
                     _conv(xs[:, 1], d32, len(x)) + _conv(xs[:, 0], c32, len(x))], axis=1)
    for ch in range(2):
        assert cases.rel_rms(y[:, ch], want[:, ch]) <= 1e-5, ch
