"""CPU: the block-level oracle (one filter_process() iteration, bfrun.c:1420-2083) against an
independent numpy linear-convolution model and analytic known-answer tests for every
bookkeeping rule the fused device path has to reproduce."""
import numpy as np
import pytest

import bforacle as bo
import cases


def _run_float(e, x, L, O, rs):
    dt = np.float32 if rs == 4 else np.float64
    outs = []
    for b in range(len(x) // L):
        st, raw = e.block(x[b * L:(b + 1) * L])
        assert st == 0
        outs.append(np.frombuffer(raw.tobytes(), dt).reshape(L, O))
    return np.concatenate(outs).astype(np.float64)


@pytest.mark.parametrize("rs,tol", [(4, 2e-6), (8, 1e-13)])
@pytest.mark.parametrize("L,N,I,O", [(64, 4, 2, 2), (32, 13, 3, 2)])      # 13: bench4's odd N
def test_crossbar_is_linear_convolution(rs, tol, L, N, I, O):
    ofmt = "FLOAT_LE" if rs == 4 else "FLOAT64_LE"
    e, irs = cases.crossbar(bo.Engine, L, N, rs, I, O, "S16_LE", ofmt)
    nblk = 2 * N + 3
    x = np.concatenate(cases.raw_blocks(5, nblk, L, I, "S16_LE"))
    y = _run_float(e, x, L, O, rs)
    xs = x.astype(np.float64) / 32768.0
    dt = np.float32 if rs == 4 else np.float64
    want = np.zeros_like(y)
    for (o, i), h in irs.items():
        want[:, o] += np.convolve(xs[:, i], h.astype(dt).astype(np.float64))[:len(x)]
    assert cases.rel_rms(y, want) <= tol


def test_scales_polarity_and_attenuation():
    """per-input scale incl. polarity flip (bench4 `0//-1`) and output scale"""
    L, N = 64, 2
    e = bo.Engine(L, N, 8, 2, 1)
    e.set_interleaved(0, "FLOAT64_LE")
    e.set_interleaved(1, "FLOAT64_LE")
    e.add_filter(in_ch=[0, 1], in_scale=[-1.0, 0.25], out_ch=[0], out_scale=[2.0], coeff=-1)
    x = np.random.default_rng(0).standard_normal((4 * L, 2))
    y = _run_float(e, x, L, 1, 8)[:, 0]
    assert np.abs(y - 2.0 * (-x[:, 0] + 0.25 * x[:, 1])).max() < 1e-12


def test_delayed_dirac_hits_the_right_partition_and_delayblocks_shift():
    L, N = 32, 6
    for part, off, dly in [(0, 0, 0), (2, 5, 0), (3, 31, 2), (1, 7, 9)]:   # 9 clamps to N-1
        e = bo.Engine(L, N, 8, 1, 1)
        e.set_interleaved(0, "FLOAT64_LE")
        e.set_interleaved(1, "FLOAT64_LE")
        taps = np.zeros(L * N)
        taps[part * L + off] = 1.0
        c = e.add_coeff(taps)
        e.add_filter(in_ch=[0], out_ch=[0], coeff=c, delayblocks=dly)
        x = np.random.default_rng(part).standard_normal((3 * N * L, 1))
        y = _run_float(e, x, L, 1, 8)[:, 0]
        d_eff = min(max(dly, 0), N - 1)
        # cblocks = min(coeff blocks, N - delay): partitions beyond it are dropped
        if part >= N - d_eff:
            assert np.abs(y).max() == 0.0
            continue
        sh = part * L + off + d_eff * L
        assert np.abs(y[sh:] - x[:len(x) - sh, 0]).max() < 1e-12
        assert sh == 0 or np.abs(y[:sh]).max() < 1e-12


def test_short_coeff_set_uses_its_own_block_count():
    L, N = 32, 8
    e = bo.Engine(L, N, 8, 1, 1)
    e.set_interleaved(0, "FLOAT64_LE")
    e.set_interleaved(1, "FLOAT64_LE")
    h = np.random.default_rng(1).standard_normal(3 * L)
    c = e.add_coeff(h, n_blocks=3)                           # `blocks: 3`
    e.add_filter(in_ch=[0], out_ch=[0], coeff=c)
    x = np.random.default_rng(2).standard_normal((12 * L, 1))
    y = _run_float(e, x, L, 1, 8)[:, 0]
    assert np.abs(y - np.convolve(x[:, 0], h)[:len(x)]).max() < 1e-11
    with pytest.raises(ValueError):
        e.add_coeff(np.zeros(9 * L), n_blocks=9)             # longer than N: rejected


def test_bench1_topology_cascade_of_diracs_is_identity():
    """bench1_config: two input filters feed two cascaded filters (from_filters), all dirac:
    both outputs = in0 + in1 (SURVEY B.5 i)"""
    L, N = 64, 8
    e = bo.Engine(L, N, 4, 2, 2)
    e.set_interleaved(0, "S24_4LE")
    e.set_interleaved(1, "S24_4LE")
    d = np.zeros(L * N)
    d[0] = 1.0
    c = e.add_coeff(d)
    f0 = e.add_filter(in_ch=[0], coeff=c)
    f1 = e.add_filter(in_ch=[1], coeff=c)
    e.add_filter(in_f=[f0, f1], out_ch=[0], coeff=c)
    e.add_filter(in_f=[f0, f1], out_ch=[1], coeff=c)
    x = np.concatenate(cases.raw_blocks(3, 12, L, 2, "S24_4LE", amplitude=0.05))
    outs = []
    for b in range(12):
        st, raw = e.block(x[b * L:(b + 1) * L])
        assert st == 0
        outs.append(raw.view(np.int32).reshape(L, 2))
    y = np.concatenate(outs).astype(np.int64)
    want = x[:, 0].astype(np.int64) + x[:, 1]
    for ch in range(2):
        assert np.abs(y[:, ch] - want).max() <= 2             # f32 rounding, 24-bit LSBs


def test_crossfade_switch_sequence():
    """coefficient switch with crossfade: the switch block is (1-w)*old + w*new with
    w = n/(L-1); afterwards ALL partitions use the new IR at once (SURVEY B.5 iv)"""
    L, N = 32, 4
    rng = np.random.default_rng(8)
    ha, hb = rng.standard_normal(L * N) * 0.1, rng.standard_normal(L * N) * 0.1
    e = bo.Engine(L, N, 8, 1, 1)
    e.set_interleaved(0, "FLOAT64_LE")
    e.set_interleaved(1, "FLOAT64_LE")
    ca, cb = e.add_coeff(ha), e.add_coeff(hb)
    f = e.add_filter(in_ch=[0], out_ch=[0], coeff=ca, crossfade=True)
    nblk = 10
    x = rng.standard_normal((nblk * L, 1))
    ya = np.convolve(x[:, 0], ha)[:nblk * L]
    yb = np.convolve(x[:, 0], hb)[:nblk * L]
    w = np.arange(L) / (L - 1.0)
    for b in range(nblk):
        if b == 5:
            e.set_coeff(f, cb)
        st, raw = e.block(x[b * L:(b + 1) * L])
        y = raw.view(np.float64)
        s = slice(b * L, (b + 1) * L)
        want = ya[s] if b < 5 else (ya[s] * (1 - w) + yb[s] * w if b == 5 else yb[s])
        assert np.abs(y - want).max() < 1e-11, b


def test_overflow_accounting_and_clipping():
    L = 64
    e = bo.Engine(L, 1, 8, 1, 1)
    e.set_interleaved(0, "FLOAT64_LE")
    e.set_interleaved(1, "S16_LE")
    e.add_filter(in_ch=[0], out_ch=[0], coeff=-1)
    x = np.zeros((L, 1))
    x[3], x[10], x[11] = 1.5, -2.0, 0.25
    st, raw = e.block(x)
    y = raw.view(np.int16)
    assert st == 0 and y[3] == 32767 and y[10] == -32768 and y[11] == 8192
    of = e.overflow(0)
    assert of.n_overflows == 2 and of.intlargest == 8192 and of.max == 32767.0
    assert of.largest == pytest.approx(2.0 * 32768 - 0.5)      # |v| after the +0.5 offset


def test_nan_and_safety_limit_are_reported():
    L = 64
    e = bo.Engine(L, 1, 4, 1, 1)
    e.set_interleaved(0, "FLOAT_LE")
    e.set_interleaved(1, "FLOAT_LE")
    e.add_filter(in_ch=[0], out_ch=[0], coeff=-1)
    e.set_safety_limit(2.0)
    x = np.zeros((L, 1), np.float32)
    assert e.block(x)[0] == 0
    x[5] = 3.0
    assert e.block(x)[0] == 2                   # reference: bf_exit (real2raw.h:32-41)
    x[5] = np.nan
    assert e.block(x)[0] == 1                   # reference: abort() (real2raw.h:24-31)


def test_virtual_channel_mapping_delay_and_mute():
    """bench4-style `mapping`: two virtual inputs fed by one physical input (own delays), two
    virtual outputs mixed into one physical output (own delays, one muted for a while) --
    bfrun.c:1509-1531 and :1938-2003; dirac filters, so the model is shifts and sums"""
    L, N = 64, 2
    e = bo.Engine(L, N, 8, 3, 3)
    e.map_channels(0, [0, 0, 1])            # virtual inputs 0,1 <- physical 0; 2 <- physical 1
    e.map_channels(1, [0, 0, 1])            # virtual outputs 0,1 -> physical 0; 2 -> physical 1
    e.set_interleaved_phys(0, "FLOAT64_LE", 2)
    e.set_interleaved_phys(1, "FLOAT64_LE", 2)
    for v in range(3):
        e.add_filter(in_ch=[v], out_ch=[v], coeff=-1)
    din, dout = [5, 70, 0], [3, 100, 0]
    for v in range(3):
        e.set_delay(0, v, din[v]); e.set_maxdelay(0, v, -1)
        e.set_delay(1, v, dout[v]); e.set_maxdelay(1, v, -1)
    nblk = 12
    rng = np.random.default_rng(5)
    x = rng.standard_normal((nblk * L, 2))
    y = []
    for b in range(nblk):
        e.set_mute(1, 1, 3 <= b < 6)        # virtual output 1 muted during blocks 3..5
        e.set_mute(0, 0, b == 8)            # virtual input 0 muted in block 8
        st, raw = e.block(x[b * L:(b + 1) * L])
        assert st == 0
        y.append(raw.view(np.float64).reshape(L, 2))
    y = np.concatenate(y)

    def shift(sig, d):
        return np.concatenate([np.zeros(d), sig[:len(sig) - d]])
    xin0 = x[:, 0].copy()
    v0_in = xin0.copy()
    # a muted input block is zeroed instead of passing through the delay line; the line itself
    # is not advanced (bfrun.c:1510-1524), so the next block continues where it stopped
    v0 = np.concatenate([shift(np.concatenate([xin0[:8 * L], xin0[9 * L:]]), din[0])[:8 * L],
                         np.zeros(L),
                         shift(np.concatenate([xin0[:8 * L], xin0[9 * L:]]), din[0])[8 * L:]])[:nblk * L]
    v1 = shift(x[:, 0], din[1])
    o0, o1 = shift(v0, dout[0]), shift(v1, dout[1])      # the output delay line always runs
    mute1 = np.zeros(nblk * L, bool)
    mute1[3 * L:6 * L] = True
    want0 = o0 + np.where(mute1, 0.0, o1)
    assert np.abs(y[:, 0] - want0).max() < 1e-12
    assert np.abs(y[:, 1] - x[:, 1]).max() < 1e-12       # the 1:1 channel is untouched
    # the overflow struct of a physical output is shared by its virtual channels (:1999-2001)
    assert e.overflow(0).astuple() == e.overflow(1).astuple()


@pytest.mark.parametrize("rs,tol", [(4, 2e-5), (8, 1e-9)])
def test_subsample_delay_is_a_fractional_shift(rs, tol):
    """sub-sample delay (delay.c:416-505): a band-limited signal through a dirac filter with
    `subdelay: k` on the input comes out delayed by sdf_length + k/100 samples"""
    L, N, half = 256, 2, 15
    e = bo.Engine(L, N, rs, 2, 2)
    e.set_interleaved(0, "FLOAT64_LE")
    e.set_interleaved(1, "FLOAT64_LE")
    e.enable_subdelay(half)
    e.set_subdelay(0, 0, 37)             # input 0: +0.37 samples
    e.set_subdelay(1, 1, -25)            # output 1: -0.25 samples
    e.add_filter(in_ch=[0], out_ch=[0], coeff=-1)
    e.add_filter(in_ch=[1], out_ch=[1], coeff=-1)
    nblk = 8
    t = np.arange(nblk * L, dtype=np.float64)
    f0 = 0.031                                       # cycles / sample: well inside the passband
    x = np.stack([np.sin(2 * np.pi * f0 * t), np.cos(2 * np.pi * f0 * t)], axis=1)
    y = np.concatenate([e.block(x[b * L:(b + 1) * L])[1].view(np.float64).reshape(L, 2) for b in range(nblk)])
    s = slice(3 * L, 7 * L)
    want0 = np.sin(2 * np.pi * f0 * (t - half - 0.37))
    want1 = np.cos(2 * np.pi * f0 * (t - half + 0.25))
    # the windowed-sinc interpolator is not exact: 31 taps with the reference's squared Kaiser(9)
    assert np.abs(y[s, 0] - want0[s]).max() < 2e-3
    assert np.abs(y[s, 1] - want1[s]).max() < 2e-3
    # subdelay 0 is a pure delay of sdf_length samples, exact to rounding
    e2 = bo.Engine(L, N, rs, 1, 1)
    e2.set_interleaved(0, "FLOAT64_LE"); e2.set_interleaved(1, "FLOAT64_LE")
    e2.enable_subdelay(half)
    e2.set_subdelay(0, 0, 0)
    e2.add_filter(in_ch=[0], out_ch=[0], coeff=-1)
    r = np.random.default_rng(1).standard_normal((nblk * L, 1))
    y2 = np.concatenate([e2.block(r[b * L:(b + 1) * L])[1].view(np.float64) for b in range(nblk)])
    assert np.abs(y2[half:] - r[:-half, 0]).max() < tol


@pytest.mark.parametrize("rs,tol", [(4, 2e-6), (8, 1e-13)])
def test_delay_cascade_crossfade_network_against_numpy(rs, tol):
    """the oracle's cross-fade (A7), cascade evaluation (A8) and block bookkeeping (A12) cannot be
    pinned by reference output here (FFTW absent): pin them by an independent float64 numpy model
    of the whole network instead; the same model checks the HIP engine in tests/test_gpu_numpy.py"""
    y, want, L, sw = cases.fade_cascade_network(bo.Engine, rs)
    for ch in range(2):
        assert cases.rel_rms(y[:, ch], want[:, ch]) <= tol, ch
    for b in (sw - 1, sw, sw + 1):
        s = slice(b * L, (b + 1) * L)
        assert cases.rel_rms(y[s, 0], want[s, 0]) <= 2 * tol, b


def test_an_input_muted_from_the_start_keeps_the_delay_it_was_configured_with():
    """the delay buffer of a shared input exists from the start (bfrun.c:1128-1166), muted or not, with
    the delay configured then; a fixed delay (maxdelay -1) is not moved by set_delay while the channel
    is muted either (delay.c:289).  The oracle once built the buffer at the first unmuted block with
    the delay of THAT moment -- the reference and the engine disagreed with it
    (tests/test_gpu_refloop.py, seed 323)."""
    L, N = 64, 2
    e = bo.Engine(L, N, 8, 2, 1)
    e.map_channels(0, [0, 0])               # both virtual inputs from physical input 0
    e.set_interleaved_phys(0, "FLOAT64_LE", 1)
    e.set_interleaved(1, "FLOAT64_LE")
    e.add_filter(in_ch=[0], out_ch=[0], coeff=-1)
    e.add_filter(in_ch=[1], out_ch=[0], coeff=-1, in_scale=[0.0])
    e.set_delay(0, 0, 100); e.set_maxdelay(0, 0, -1)
    e.set_mute(0, 0, 1)
    nblk = 10
    x = np.random.default_rng(9).standard_normal((nblk * L, 1))
    y = []
    for b in range(nblk):
        if b == 2:
            e.set_delay(0, 0, 30)           # while muted, on a fixed delay: nothing
        if b == 4:
            e.set_mute(0, 0, 0)
        st, raw = e.block(x[b * L:(b + 1) * L])
        assert st == 0
        y.append(raw.view(np.float64).reshape(L))
    y = np.concatenate(y)
    heard = x[4 * L:, 0]                    # what the delay line has seen: the blocks since the unmute
    want = np.concatenate([np.zeros(4 * L), np.zeros(100), heard[:len(heard) - 100]])
    assert np.abs(y - want).max() < 1e-12


def test_fixed_delays_beside_a_subsample_filter_in_the_oracle():
    """maxdelay -1 on channels that share a physical one, beside a sub-sample filter on the same side: the
    reference's set-up overruns its delay buffer there (bfrun.c:1152-1162 + delay.c:357-374);
    the oracle's (and the engine's: tests/test_gpu_features.py) defined answer is delay + sdf_length,
    fixed -- an impulse comes out where that says"""
    L, N, half = 64, 2, 15
    e = bo.Engine(L, N, 4, 3, 3)
    e.map_channels(0, [0, 0, 1])
    e.map_channels(1, [0, 0, 1])
    e.set_interleaved_phys(0, "FLOAT_LE", 2)
    e.set_interleaved_phys(1, "FLOAT_LE", 2)
    e.enable_subdelay(half)
    e.set_subdelay(0, 0, 0)
    e.set_subdelay(1, 0, 0)
    for io in range(2):
        e.set_maxdelay(io, 1, -1); e.set_delay(io, 1, 700 if io == 0 else 20)
        e.set_maxdelay(io, 0, 40); e.set_delay(io, 0, 90)
    e.add_filter(in_ch=[1], out_ch=[1], coeff=-1)
    e.add_filter(in_ch=[0], out_ch=[0], coeff=-1, in_scale=[0.0])
    e.add_filter(in_ch=[2], out_ch=[2], coeff=-1)
    x = np.zeros((16 * L, 2), np.float32)
    x[5, 0] = 1.0
    got = []
    for b in range(16):
        if b == 3:
            e.set_delay(0, 1, 10)               # refused: the delay is fixed
        st, raw = e.block(x[b * L:(b + 1) * L])
        assert st == 0
        got.append(raw.view(np.float32).reshape(L, 2)[:, 0])
    y = np.concatenate(got)
    assert int(np.argmax(np.abs(y))) == 5 + (700 + half) + (20 + half) and abs(float(y.max()) - 1.0) < 1e-5
