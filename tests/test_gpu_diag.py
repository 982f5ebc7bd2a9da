"""GPU: one-to-one plans (every output fed by one single-term filter: massive_config, BASELINE
configs[3]) run the MAC as mac_diag_kernel -- one workgroup per (part, output) walking whole
spectra, parallel over parts of the partition axis -- instead of the crossbar kernel (kernels.h).
Checked against the oracle, against the crossbar kernel on the same plan, and for the properties the
rest of the engine relies on (warm-up, powersave, graph replay, shards)."""
import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu
FLT = {4: ("FLOAT_LE", np.float32, 1e-5), 8: ("FLOAT64_LE", np.float64, 1e-12)}


def _one_to_one(cls, L, N, rs, C, seed=3, powersave=0.0, shard=None):
    """C channels, filter c: input (c * 7) % C -> output c (C coprime to 7: every input feeds one filter), ragged set lengths, delays, scales, one dirac-free"""
    rng = np.random.default_rng(seed)
    e = cls(L, N, rs, C, C)
    e.set_interleaved(0, "S24_4LE")
    e.set_interleaved(1, FLT[rs][0])
    if powersave:
        e.set_powersave(powersave)
    for c in range(C):
        nb = int(rng.integers(1, N + 1))
        h = cases.make_ir(rng, L * nb - int(rng.integers(0, L // 2)), 1).astype(FLT[rs][1])
        f = e.add_filter(in_ch=[(c * 7) % C], in_scale=[float(rng.choice([1.0, -0.5, 0.25]))], out_ch=[c],
                         out_scale=[float(rng.choice([1.0, 2.0]))], coeff=e.add_coeff(h), delayblocks=int(rng.integers(0, 3)))
        if shard is not None:
            e.set_filter_active(f, c % 2 == shard)
    if hasattr(e, "finalize"):
        e.finalize()
    return e


@pytest.mark.parametrize("L,N,rs,C,diag", [(256, 5, 4, 11, True), (1024, 3, 8, 9, True), (8192, 4, 4, 20, True),
                                           (4096, 6, 8, 5, True), (8192, 3, 8, 6, True), (64, 2, 4, 3, True)])
def test_one_to_one_plan_against_oracle(hip, monkeypatch, L, N, rs, C, diag):
    ge = _one_to_one(hip.Engine, L, N, rs, C)
    assert ge.uses_diag_mac == diag           # (float64 at L = 8192: 32 tiles, two workgroups of 16 per job)
    oe = _one_to_one(bo.Engine, L, N, rs, C)
    monkeypatch.setenv("BFHIP_MAC_DIAG", "0")
    xe = _one_to_one(hip.Engine, L, N, rs, C)
    monkeypatch.delenv("BFHIP_MAC_DIAG")
    assert not xe.uses_diag_mac
    _, dt, tol = FLT[rs]
    for blk in cases.raw_blocks(21, N + 4, L, C, "S24_4LE"):
        gs, g = ge.block(blk)
        _, x = xe.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        assert cases.rel_rms(np.frombuffer(g.tobytes(), dt), np.frombuffer(o.tobytes(), dt)) <= tol
        assert cases.rel_rms(np.frombuffer(g.tobytes(), dt), np.frombuffer(x.tobytes(), dt)) <= tol


@pytest.mark.parametrize("split", ["1", "2", "3", "4"])
def test_partition_split_of_the_diag_kernel(hip, monkeypatch, split):
    """BFHIP_DIAG_SPLIT parts of the partition axis per entry (= partial spectra per output)"""
    monkeypatch.setenv("BFHIP_DIAG_SPLIT", split)
    ge = _one_to_one(hip.Engine, 2048, 7, 4, 13)
    monkeypatch.delenv("BFHIP_DIAG_SPLIT")
    assert ge.uses_diag_mac
    oe = _one_to_one(bo.Engine, 2048, 7, 4, 13)
    for blk in cases.raw_blocks(5, 11, 2048, 13, "S24_4LE"):
        _, g = ge.block(blk)
        _, o = oe.block(blk)
        assert cases.rel_rms(np.frombuffer(g.tobytes(), np.float32), np.frombuffer(o.tobytes(), np.float32)) <= 1e-5


def test_an_output_fed_twice_is_not_a_diag_plan(hip):
    e = hip.Engine(256, 2, 4, 3, 3)
    e.set_interleaved(0, "S24_4LE")
    e.set_interleaved(1, "FLOAT_LE")
    c = e.add_coeff(np.ones(256, np.float32))
    for i, o in ((0, 0), (1, 1), (2, 1)):
        e.add_filter(in_ch=[i], out_ch=[o], coeff=c)
    e.finalize()
    assert not e.uses_diag_mac
    d = hip.Engine(256, 2, 4, 3, 3)               # ... and a dirac term is no coefficient term
    d.set_interleaved(0, "S24_4LE")
    d.set_interleaved(1, "FLOAT_LE")
    d.add_filter(in_ch=[0], out_ch=[0], coeff=-1)
    d.finalize()
    assert not d.uses_diag_mac


def test_diag_plan_warm_up_powersave_and_graph_replay(hip):
    """the properties the crossbar kernel has: a prewarmed engine computes the same bits (the partition
    loop is cut by `age`, not reordered), silent inputs are skipped without changing a sample, and a
    replayed graph (block counter from device memory) equals the plain launches"""
    L, N, C = 1024, 4, 10
    blocks = cases.raw_blocks(31, N + 5, L, C, "S24_4LE")
    for b in (2, 3, 4, 5, 6, 7):
        blocks[b][:, 3] = 0                       # channel 3 falls silent for a whole filter length
    a = _one_to_one(hip.Engine, L, N, 4, C)
    w = _one_to_one(hip.Engine, L, N, 4, C)
    w.prewarm()
    ps = _one_to_one(hip.Engine, L, N, 4, C, powersave=1.0)
    rt = _one_to_one(hip.Engine, L, N, 4, C)
    assert a.uses_diag_mac and ps.uses_diag_mac and rt.uses_diag_mac
    rt.rt_begin(0)
    for blk in blocks:
        _, g = a.block(blk)
        assert np.array_equal(w.block(blk)[1], g)
        assert np.array_equal(ps.block(blk)[1], g)
        assert np.array_equal(rt.rt_block(blk)[1], g)
    assert rt.rt_stats()["graph"] > 0
    rt.rt_end()


def test_diag_plan_in_shards_is_bit_identical(hip):
    L, N, C = 512, 6, 15
    whole = _one_to_one(hip.Engine, L, N, 4, C)
    shards = [_one_to_one(hip.Engine, L, N, 4, C, shard=k) for k in range(2)]
    assert whole.uses_diag_mac and all(s.uses_diag_mac for s in shards)
    for blk in cases.raw_blocks(8, N + 4, L, C, "S24_4LE"):
        _, w = whole.block(blk)
        shared = np.zeros(w.size, np.uint8)
        for s in shards:
            s.block(blk, out=shared)
        assert np.array_equal(shared, w)
