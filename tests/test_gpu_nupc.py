"""GPU: the non-uniform partitioned convolver (include/bfhip_nupc.h; BASELINE.json configs[4]:
"non-uniform partition sizes, low-latency first block").  The reference only has uniform
partitions, so parity is "the same linear convolution as a uniform run": the oracle's uniform
engine processes the same stream with the same impulse responses and the two output streams
must agree sample for sample (float32 1e-5 / float64 1e-12 relative RMS)."""
import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu
TOL = {4: 1e-5, 8: 1e-12}
FF = {4: "FLOAT_LE", 8: "FLOAT64_LE"}


def _run_pair(hip, rs, seg_len, seg_blk, n_in, n_out, n_frames, oracle_L, infmt="S24_4LE", taps=None, seed=0):
    nu = hip.Nupc(seg_len, seg_blk, rs, n_in, n_out)
    nu.set_interleaved(0, infmt)
    nu.set_interleaved(1, FF[rs])
    taps = nu.taps if taps is None else taps
    assert taps <= nu.taps
    oN = -(-taps // oracle_L)
    oe = bo.Engine(oracle_L, oN, rs, n_in, n_out)
    oe.set_interleaved(0, infmt)
    oe.set_interleaved(1, FF[rs])
    dt = np.float32 if rs == 4 else np.float64
    for o in range(n_out):
        for i in range(n_in):
            h = cases.make_ir(np.random.default_rng(seed + 10 * o + i), taps, n_in).astype(dt)
            nu.add_filter(i, o, h, in_scale=0.5 if i else 1.0, out_scale=-1.0 if o else 1.0)
            oe.add_filter(in_ch=[i], out_ch=[o], coeff=oe.add_coeff(h),
                          in_scale=[0.5 if i else 1.0], out_scale=[-1.0 if o else 1.0])
    nu.finalize()
    L0 = nu.L0
    assert n_frames % oracle_L == 0
    x = np.concatenate(cases.raw_blocks(seed + 5, n_frames // oracle_L, oracle_L, n_in, infmt))
    got = []
    for b in range(n_frames // L0):
        st, raw = nu.block(x[b * L0:(b + 1) * L0])
        assert st == 0
        got.append(np.frombuffer(raw.tobytes(), dt).reshape(L0, n_out))
    want = []
    for b in range(n_frames // oracle_L):
        st, raw = oe.block(x[b * oracle_L:(b + 1) * oracle_L])
        assert st == 0
        want.append(np.frombuffer(raw.tobytes(), dt).reshape(oracle_L, n_out))
    return np.concatenate(got).astype(np.float64), np.concatenate(want).astype(np.float64), nu


@pytest.mark.parametrize("rs", [4, 8])
def test_small_schedule_equals_uniform_convolution(hip, rs):
    """2x64, 2x128, 2x256, 6x512 = 3968 taps, latency 64 frames; oracle: uniform 64 x 62"""
    got, want, nu = _run_pair(hip, rs, [64, 128, 256, 512], [2, 2, 2, 6], 2, 2, 64 * 96, 64)
    assert nu.L0 == 64 and nu.taps == 3968
    # block by block: every 64-frame block is right on time, including the first ones
    for b in range(0, len(got), 64):
        w = want[b:b + 64]
        if np.abs(w).max() > 0:
            assert cases.rel_rms(got[b:b + 64], w) <= TOL[rs] * 3, b
    assert cases.rel_rms(got, want) <= TOL[rs]


def test_schedule_must_be_causal(hip):
    with pytest.raises(hip.BfhipError, match="not be ready in time"):
        hip.Nupc([64, 1024], [2, 4], 4, 1, 1)          # 1024-segment would start at tap 128 < 960
    with pytest.raises(hip.BfhipError, match="ascend"):
        hip.Nupc([128, 64], [2, 2], 4, 1, 1)


def test_room_correction_million_taps_float64_low_latency(hip):
    """BASELINE configs[4]: 2-in/2-out, 1048576-tap filters, float64, non-uniform partitions:
    2 x 64 ... 2 x 4096 doubling, then 127 x 8192 = 1056640 taps, I/O latency 64 frames instead
    of 8192.  Checked against the oracle's uniform 8192 x 129 run over 3 x 8192 frames."""
    seg_len = [64 << k for k in range(7)] + [8192]
    seg_blk = [2] * 7 + [127]
    got, want, nu = _run_pair(hip, 8, seg_len, seg_blk, 2, 2, 3 * 8192, 8192, infmt="FLOAT64_LE",
                              taps=1048576, seed=40)
    assert nu.L0 == 64 and nu.taps >= 1048576
    assert cases.rel_rms(got, want) <= 1e-12
    assert cases.rel_rms(got[:64], want[:64]) <= 1e-11           # the very first 64-frame block
