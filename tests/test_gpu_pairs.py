"""GPU: two consecutive blocks with ONE pass over the coefficients (bfhip_engine_block_pair_dev,
mac_xbar2_kernel; exploratory -- VERDICT r2 item 8).  Whatever path a pair takes -- the paired kernel
for a warm uniform crossbar, two single blocks for everything else -- both output buffers hold the
bits two bfhip_engine_block_dev calls leave."""
import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _crossbar(cls, L, N, rs, I, O, pairs=False, delays=False, short=False, powersave=0.0):
    e = cls(L, N, rs, I, O)
    e.set_interleaved(0, "S24_4LE")
    e.set_interleaved(1, "FLOAT_LE" if rs == 4 else "FLOAT64_LE")
    if powersave:
        e.set_powersave(powersave)
    if pairs:
        e.enable_pairs(True)
    dt = np.float32 if rs == 4 else np.float64
    for o in range(O):
        for i in range(I):
            rng = np.random.default_rng(900 + o * I + i)
            taps = L * N if not (short and (o + i) % 5 == 0) else L * (N - 1)
            c = e.add_coeff(cases.make_ir(rng, taps, I).astype(dt))
            e.add_filter(in_ch=[i], in_scale=[float(rng.choice([1.0, -0.5]))], out_ch=[o], coeff=c,
                         delayblocks=int(i % 2) if delays else 0)
    alt = e.add_coeff(cases.make_ir(np.random.default_rng(5), L * N, I).astype(dt))
    if hasattr(e, "finalize"):
        e.finalize()
    return e, alt


@pytest.mark.parametrize("rs,stream,delays,short", [(4, "2", False, False), (4, "0", True, False), (8, "2", True, False),
                                                    (4, "2", False, True)])
def test_block_pairs_equal_two_single_blocks(hip, monkeypatch, rs, stream, delays, short):
    torch = _torch()
    dev = torch.device("cuda", 0)
    L, N, I, O = 1024, 5, 8, 16
    monkeypatch.setenv("BFHIP_COEFF_STREAM", stream)
    monkeypatch.setenv("BFHIP_MAC_TARGET_WGS", "16")
    pe, palt = _crossbar(hip.Engine, L, N, rs, I, O, pairs=True, delays=delays, short=short)
    se, salt = _crossbar(hip.Engine, L, N, rs, I, O, delays=delays, short=short)
    oe, oalt = _crossbar(bo.Engine, L, N, rs, I, O, delays=delays, short=short)
    dt, odt, tol = (np.float32, torch.float32, 1e-5) if rs == 4 else (np.float64, torch.float64, 1e-12)
    blocks = cases.raw_blocks(77, 4 * N + 2, L, I, "S24_4LE")
    srcs = [torch.from_numpy(b).to(dev) for b in blocks]
    outs = [torch.zeros(L, O, dtype=odt, device=dev) for _ in blocks]
    want = []
    for k, blk in enumerate(blocks):
        if k == 2 * N + 2:                       # a coefficient switch between two pairs: plan rebuild
            for e, c in ((pe, palt), (se, salt), (oe, oalt)):
                e.set_coeff(3, c)
        _, w = se.block(blk)
        want.append(np.frombuffer(w.tobytes(), dt).reshape(L, O))
        _, o = oe.block(blk)
        assert cases.rel_rms(want[-1], np.frombuffer(o.tobytes(), dt).reshape(L, O)) <= tol
        if k % 2 == 1:
            pe.block_pair_dev(srcs[k - 1], outs[k - 1], srcs[k], outs[k])
    assert pe.sync() == 0
    for k in range(len(blocks)):
        assert np.array_equal(outs[k].cpu().numpy(), want[k]), k
    # the first N blocks (rings not full) went through single blocks; afterwards the paired kernel ran
    # -- unless the plan is no uniform crossbar (short sets), where every pair is two single blocks
    assert pe.pair_launches == (0 if short else (len(blocks) // 2 - (N + 1) // 2))


def test_pairs_with_powersave_and_single_blocks_in_between(hip):
    torch = _torch()
    dev = torch.device("cuda", 0)
    L, N, I, O = 2048, 3, 4, 8
    pe, _ = _crossbar(hip.Engine, L, N, 4, I, O, pairs=True, powersave=1.0)
    se, _ = _crossbar(hip.Engine, L, N, 4, I, O, powersave=1.0)
    blocks = cases.raw_blocks(3, 14, L, I, "S24_4LE")
    for b in (5, 6, 7, 8, 9):
        blocks[b][:, 2] = 0                      # an input silent for more than a filter length
    srcs = [torch.from_numpy(b).to(dev) for b in blocks]
    outs = [torch.zeros(L, O, dtype=torch.float32, device=dev) for _ in blocks]
    order = [(0,), (1, 2), (3,), (4, 5), (6, 7), (8,), (9, 10), (11, 12), (13,)]       # pairs and single blocks mixed
    for grp in order:
        if len(grp) == 1:
            pe.block_dev(srcs[grp[0]], outs[grp[0]])
        else:
            pe.block_pair_dev(srcs[grp[0]], outs[grp[0]], srcs[grp[1]], outs[grp[1]])
    assert pe.sync() == 0 and pe.pair_launches >= 3
    for k, blk in enumerate(blocks):
        _, w = se.block(blk)
        assert np.array_equal(outs[k].cpu().numpy(), np.frombuffer(w.tobytes(), np.float32).reshape(L, O)), k


def test_pairs_need_the_switch_before_finalize(hip):
    e, _ = _crossbar(hip.Engine, 256, 2, 4, 2, 2)
    with pytest.raises(hip.BfhipError, match="after finalize"):
        e.enable_pairs(True)
