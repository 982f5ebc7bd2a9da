"""GPU: the block counter's wrap.  Ring slots are `counter mod depth`; a depth that is not a power
of two -- N = 13 (bench4_config), or the spare slot the pipelined schedules add (N + 1) -- would
jump when an unsigned counter wraps at 2^32 (66 days of 64-frame blocks).  The engine wraps its
counter by a multiple of every ring depth it uses instead (kernels.h, BlockState).  Here the wrap
is pulled in to a few ring periods (BFHIP_TEST_WRAP_PERIODS) and crossed several times, on the host
path and inside a replayed HIP graph (where the device advances the counter), with shared rings,
private rings (an N-way input mix), delays and a cascade; the oracle, whose own counter never
wraps in these runs, is the reference."""
import numpy as np
import pytest

import bforacle as bo
import cases
from test_gpu_features import _ir, _spec

pytestmark = pytest.mark.gpu


def _network(L, N, rs):
    I, O = 3, 2
    coeffs = [(_ir(60 + k, L * N, I), 1.0, 0) for k in range(5)]
    filters = [
        dict(in_ch=[0], out_ch=[0], coeff=0, delayblocks=1),
        dict(in_ch=[1], out_ch=[1], coeff=1),
        dict(in_ch=[0, 2], in_scale=[0.5, -0.25], out_ch=[0], coeff=2, delayblocks=N - 1),    # private ring, depth N
        dict(in_ch=[2], out_ch=[], coeff=3),                                                    # cascade source
        dict(in_ch=[1], in_f=[3], in_fscale=[0.5], out_ch=[1], coeff=4, delayblocks=2),
    ]
    return _spec(L, N, rs, I, O, filters, coeffs)


@pytest.mark.parametrize("L,N,overlap", [(256, 3, "0"), (256, 13, "0"), (1024, 3, "1"), (4096, 3, "1")])
def test_counter_wrap_keeps_every_ring_aligned(hip, monkeypatch, L, N, overlap):
    spec = _network(L, N, 4)
    monkeypatch.setenv("BFHIP_TEST_WRAP_PERIODS", "2")
    monkeypatch.setenv("BFHIP_OVERLAP", overlap)
    monkeypatch.setenv("BFHIP_DEFER", "0")
    ge = cases.build(hip.Engine, spec)
    for k in ("BFHIP_TEST_WRAP_PERIODS", "BFHIP_OVERLAP", "BFHIP_DEFER"):
        monkeypatch.delenv(k)
    oe = cases.build(bo.Engine, spec)
    R = ge.ring_depth
    assert R == (N + 1 if overlap == "1" else N)
    period = N if R == N else N * R
    n_blocks = 5 * period + 2 * R + 3                  # the wrap (every 2 periods) is crossed twice at least
    seen, wrapped = [], 0
    for k, blk in enumerate(cases.raw_blocks(8, n_blocks, L, 3, "S24_4LE")):
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        err = cases.rel_rms(cases.samples(g, spec["outfmt"]), cases.samples(o, spec["outfmt"]))
        assert err <= 1e-5, (k, err)
        c = ge.blockcounter
        wrapped += bool(seen and c < seen[-1])
        seen.append(c)
    assert wrapped >= 2 and max(seen) < 2 * period + 2 * R      # it did wrap, and where it should


def test_counter_wrap_inside_a_replayed_graph(hip, monkeypatch):
    """real-time mode: the tail kernel advances the counter on the device (BlockState)"""
    L, N = 256, 5
    spec = _network(L, N, 4)
    monkeypatch.setenv("BFHIP_TEST_WRAP_PERIODS", "1")
    ge = cases.build(hip.Engine, spec)
    monkeypatch.delenv("BFHIP_TEST_WRAP_PERIODS")
    oe = cases.build(bo.Engine, spec)
    ge.rt_begin(0)
    n_blocks = 4 * N * (N + 1) + 7
    for k, blk in enumerate(cases.raw_blocks(9, n_blocks, L, 3, "S24_4LE")):
        gs, g = ge.rt_block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        err = cases.rel_rms(cases.samples(g, spec["outfmt"]), cases.samples(o, spec["outfmt"]))
        assert err <= 1e-5, (k, err)
    st = ge.rt_stats()
    assert st["graph"] >= n_blocks - 3             # the blocks really were replayed
    ge.rt_end()


def test_a_wrap_that_is_not_a_multiple_of_the_ring_depth_is_what_goes_wrong(hip, monkeypatch):
    """the sensitivity of the tests above: wrap by one block more than a ring period -- what
    2^32 is to a depth of 13 -- and the history the filters see is misaligned from then on"""
    L, N = 256, 13
    spec = _network(L, N, 4)
    monkeypatch.setenv("BFHIP_TEST_WRAP_PERIODS", "1")
    monkeypatch.setenv("BFHIP_TEST_WRAP_SKEW", "1")
    monkeypatch.setenv("BFHIP_OVERLAP", "0")
    monkeypatch.setenv("BFHIP_DEFER", "0")
    ge = cases.build(hip.Engine, spec)
    for k in ("BFHIP_TEST_WRAP_PERIODS", "BFHIP_TEST_WRAP_SKEW", "BFHIP_OVERLAP", "BFHIP_DEFER"):
        monkeypatch.delenv(k)
    oe = cases.build(bo.Engine, spec)
    errs = []
    for blk in cases.raw_blocks(8, 4 * N + 5, L, 3, "S24_4LE"):
        _, g = ge.block(blk)
        _, o = oe.block(blk)
        errs.append(cases.rel_rms(cases.samples(g, spec["outfmt"]), cases.samples(o, spec["outfmt"])))
    first_wrap = N + 2 * N + 1                       # wrap_at = period + 1 + 2 R blocks in
    assert max(errs[:first_wrap - 1]) <= 1e-5 and max(errs[first_wrap:]) > 1e-2, errs
