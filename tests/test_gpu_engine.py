"""GPU parity: the HIP engine through the C ABI vs the CPU oracle on the same seeded inputs.
float32: <= 1e-5 relative RMS on the raw output samples (north_star); overflow / peak
counters and block bookkeeping bit-exact."""
import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-5


def _as_samples(raw, fmt):
    return np.frombuffer(raw.tobytes(), cases.RAW_NP[fmt]).astype(np.float64)


@pytest.mark.parametrize("L,N,I,O", [(64, 4, 2, 2), (256, 8, 3, 5), (8192, 8, 8, 8),
                                     (256, 37, 1, 8), (128, 5, 3, 16), (512, 2, 9, 24)])   # all-crossbar plans: the pipelined MAC
def test_crossbar_matches_oracle_f32(hip, L, N, I, O):
    """S24_4LE in, FLOAT_LE out: the float32 tolerance of north_star on unquantised samples"""
    ifmt, ofmt = "S24_4LE", "FLOAT_LE"
    ge, _ = cases.crossbar(hip.Engine, L, N, 4, I, O, ifmt, ofmt)
    oe, _ = cases.crossbar(bo.Engine, L, N, 4, I, O, ifmt, ofmt)
    blocks = cases.raw_blocks(1234, N + 3, L, I, ifmt)
    gs, gout = cases.run(ge, blocks)
    os_, oout = cases.run(oe, blocks)
    assert gs == os_ == [0] * len(blocks)
    for b, (g, o) in enumerate(zip(gout, oout)):
        err = cases.rel_rms(_as_samples(g, ofmt), _as_samples(o, ofmt))
        assert err <= TOL_F32, (b, err)
    for ch in range(O):
        g, o = ge.overflow(ch), oe.overflow(ch)
        assert g.n_overflows == o.n_overflows == 0
        assert (g.max, g.intlargest) == (o.max, o.intlargest) == (1.0, 0)
        assert abs(g.largest - o.largest) <= 1e-5 * o.largest
    assert ge.blockcounter == len(blocks)


@pytest.mark.parametrize("L,N,I,O", [(64, 4, 2, 2), (8192, 8, 8, 8)])
def test_crossbar_quantised_s24(hip, L, N, I, O):
    """S24_4LE out: after the mid-tread requantiser the two paths may differ by one LSB on
    the few samples whose float value sits within float32 rounding of a .5 boundary"""
    fmt = "S24_4LE"
    ge, _ = cases.crossbar(hip.Engine, L, N, 4, I, O, fmt, fmt)
    oe, _ = cases.crossbar(bo.Engine, L, N, 4, I, O, fmt, fmt)
    blocks = cases.raw_blocks(1234, N + 3, L, I, fmt)
    gs, gout = cases.run(ge, blocks)
    os_, oout = cases.run(oe, blocks)
    assert gs == os_ == [0] * len(blocks)
    for b, (g, o) in enumerate(zip(gout, oout)):
        gsamp, osamp = _as_samples(g, fmt), _as_samples(o, fmt)
        d = np.abs(gsamp - osamp)
        assert d.max() <= 1.0, b
        assert (d > 0).mean() < 0.05, (b, (d > 0).mean())
    for ch in range(O):
        g, o = ge.overflow(ch), oe.overflow(ch)
        assert g.n_overflows == o.n_overflows == 0
        assert g.max == o.max
        assert abs(g.intlargest - o.intlargest) <= 1
    assert ge.blockcounter == len(blocks)


def test_input_spectrum_matches_numpy(hip):
    """K1 alone: ring slot = rfft of [previous block | this block] (FFTW R2HC definition)"""
    L, N, I = 1024, 2, 2
    ge, _ = cases.crossbar(hip.Engine, L, N, 4, I, 1, "FLOAT_LE", "FLOAT_LE")
    blocks = cases.raw_blocks(7, 3, L, I, "FLOAT_LE")
    prev = np.zeros((L, I), np.float32)
    for t, b in enumerate(blocks):
        ge.block(b)
        for ch in range(I):
            X = np.fft.rfft(np.concatenate([prev[:, ch], b[:, ch]]).astype(np.float64))
            z = ge.ring_slot(ch, t % ge.ring_depth).astype(np.complex128)
            got = np.concatenate([[z[0].real], z[1:], [z[0].imag]])
            assert np.abs(got - X).max() / np.abs(X).max() < 2e-6
        prev = b
