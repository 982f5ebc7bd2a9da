"""GPU: partition lengths above the LDS limit (brutefir_amd/csrc/bigfft.h).  The reference's
stock configuration is one partition of 65536 taps (`filter_length: 65536;`, bfconf.c:197,
bench3_config); 16384 ... 65536 run the FFT-bearing steps as multi-kernel sequences over global
memory.  Same oracle, same tolerances as the short lengths."""
import numpy as np
import pytest

import bforacle as bo
import cases
from test_gpu_features import _compare, _ir, _spec, FLOATFMT, TOL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L,N,rs", [(16384, 2, 4), (32768, 1, 4), (65536, 1, 4), (65536, 2, 8), (16384, 3, 8),
                                    (131072, 1, 4), (262144, 2, 8), (524288, 1, 4), (1048576, 1, 8)])
def test_crossbar_long_partitions(hip, L, N, rs):
    """bench3_config's shape: few channels, one or two very long partitions"""
    I, O = 2, 2
    coeffs = [(_ir(200 + k, L * N, I), 1.0, 0) for k in range(I * O)]
    filters = [dict(in_ch=[i], out_ch=[o], coeff=o * I + i) for o in range(O) for i in range(I)]
    _compare(hip, _spec(L, N, rs, I, O, filters, coeffs), 2 * N + 2)


@pytest.mark.parametrize("rs", [4, 8])
def test_every_filter_feature_at_16384(hip, rs):
    """cascade with a mixed channel + filter input, dirac, delays (incl. clamp), short sets,
    scales, and a cross-faded / hard coefficient switch sequence, at L = 16384"""
    L, N = 16384, 3
    coeffs = [(_ir(210 + k, L * N, 2), 1.0, 0) for k in range(3)] + [(_ir(214, L, 2), 0.5, 1)]
    filters = [
        dict(in_ch=[0], coeff=0, out_ch=[2], crossfade=True),                              # 0
        dict(in_ch=[1], coeff=1, delayblocks=1),                                            # 1
        dict(in_f=[0, 1], in_fscale=[1.0, -0.5], out_ch=[0], coeff=2, crossfade=True),      # 2
        dict(in_ch=[0, 1], in_scale=[0.5, 0.25], in_f=[1], out_ch=[1], coeff=3),            # 3
        dict(in_ch=[1], out_ch=[0, 1], out_scale=[0.25, -1.0], coeff=-1, delayblocks=7),    # 4: dirac, clamped
    ]
    plan = {2: [(0, 1)], 3: [(2, -1)], 4: [(0, 0), (2, 2)], 5: [(2, 0)]}

    def control(b, eng):
        for f, c in plan.get(b, []):
            eng.set_coeff(f, c)
    tol = 5e-5 if rs == 4 else 1e-11
    _compare(hip, _spec(L, N, rs, 2, 3, filters, coeffs), 8, control=control, tol=tol)


def test_quantised_and_dithered_outputs_at_32768(hip):
    """integer output (overflow counters equal) and HP-TPDF dithered S16 in float64 (bit exact
    like the short lengths: one dither chain per channel, same table walk)"""
    L, N = 32768, 1
    coeffs = [(_ir(220, L * N), 1.0, 0), (_ir(221, L * N), 1.0, 0)]
    filters = [dict(in_ch=[0], out_ch=[0], coeff=0), dict(in_ch=[1], out_ch=[1], coeff=1, out_scale=[3.0])]
    _compare(hip, _spec(L, N, 4, 2, 2, filters, coeffs, outfmt="S24_4LE"), 4)

    def mk(cls):
        e = cls(L, N, 8, 2, 2)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "S16_LE")
        for t, s_, nb in coeffs:
            e.add_coeff(t, s_, nb)
        for f in filters:
            e.add_filter(**f)
        e.enable_dither([0, 1], 44100, 0)
        if hasattr(e, "finalize"):
            e.finalize()
        return e
    ge, oe = mk(hip.Engine), mk(bo.Engine)
    for b, blk in enumerate(cases.raw_blocks(7, 3, L, 2, "S24_4LE", amplitude=0.3)):
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_
        assert np.array_equal(g, o), b


def test_runtime_coefficient_update_at_65536(hip):
    """bfhip_engine_update_coeff_block through the big coefficient-prep sequence: with one
    partition the output of a block depends on that block and the one before only, so a fresh
    oracle that had the new taps all along must agree from the block of the update on"""
    L, N = 65536, 1
    h0, h1 = _ir(230, L), _ir(231, L)
    spec = _spec(L, N, 4, 1, 1, [dict(in_ch=[0], out_ch=[0], coeff=0)], [(h0, 1.0, 0)])
    ge = cases.build(hip.Engine, spec)
    oe = cases.build(bo.Engine, spec)
    blocks = cases.raw_blocks(9, 4, L, 1, "S24_4LE")
    for b, blk in enumerate(blocks):
        if b == 2:
            ge.update_coeff_block(0, 0, h1.astype(np.float32))
            oe = cases.build(bo.Engine, dict(spec, coeffs=[(h1, 1.0, 0)]))
            oe.block(blocks[1])
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        assert cases.rel_rms(cases.samples(g, "FLOAT_LE"), cases.samples(o, "FLOAT_LE")) <= TOL[4], b


@pytest.mark.parametrize("L", [4, 8, 16, 32])
@pytest.mark.parametrize("rs", [4, 8])
def test_tiny_partitions(hip, L, rs):
    """the other end of the range: bfconf.c:1512-1514 allows any power of two with L * N >= 4"""
    N, I, O = 5, 2, 3
    coeffs = [(_ir(240 + k, L * N, I), 1.0, 0) for k in range(4)]
    filters = [dict(in_ch=[0], out_ch=[0], coeff=0), dict(in_ch=[1], out_ch=[1], coeff=1, delayblocks=2),
               dict(in_ch=[0, 1], in_scale=[0.5, -0.5], out_ch=[2], coeff=2),
               dict(in_f=[0], in_ch=[1], out_ch=[2], coeff=3), dict(in_ch=[0], out_ch=[1], coeff=-1)]
    _compare(hip, _spec(L, N, rs, I, O, filters, coeffs), 3 * N)


def test_length_limits(hip):
    with pytest.raises(hip.BfhipError, match="Invalid length"):
        hip.Engine(2097152, 1, 4, 1, 1)
    with pytest.raises(hip.BfhipError, match="Invalid length"):
        hip.Engine(2, 4, 4, 1, 1)
    with pytest.raises(hip.BfhipError, match="Invalid length"):
        hip.Engine(96, 4, 4, 1, 1)
