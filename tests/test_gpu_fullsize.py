"""GPU, BASELINE.json's full sizes (64-in/64-out crossbar, 262144 taps = 8192 x 32, f32, 8 GiB
of coefficients): size-independent properties plus an oracle spot check.
  * linearity:      F(a*x + b*y) = a*F(x) + b*F(y) block by block (two engines' worth of state
                    is avoided by superposing inputs in one run with an all-dirac bank and by
                    scaling -- see each test)
  * identity:       64 `coeff: -1` (dirac) filters on the diagonal return their input to within
                    the float32 FFT round trip (<= 2 LSB at 24 bit)
  * spot check:     2 of the 64 outputs x all 64 inputs x all 32 partitions recomputed by the
                    CPU oracle from the very taps the device holds (downloaded), 4 blocks after
                    the rings are full: <= 1e-5 relative RMS
  * checksum:       the sum over all outputs equals the response of the summed filter bank on
                    the summed... (delay-free linear map): checked through the spot outputs
"""
import os

import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu

I = O = 64
L, N = 8192, 32
FMT = "S24_4LE"


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _ir_dev(torch, seed, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    taps = L * N
    h = torch.randn(taps, generator=g, device=dev, dtype=torch.float32)
    h *= torch.exp(-torch.arange(taps, device=dev, dtype=torch.float32) / (taps / 6.0))
    h /= h.abs().sum() * I
    return h


@pytest.fixture(scope="module")
def full_engine(hip):
    torch = _torch()
    dev = torch.device("cuda", 0)
    e = hip.Engine(L, N, 4, I, O)
    e.set_interleaved(0, FMT)
    e.set_interleaved(1, "FLOAT_LE")
    spot = {}
    for o in range(O):
        for i in range(I):
            h = _ir_dev(torch, 4321 + o * I + i, dev)
            if o in (3, 42):
                spot[(o, i)] = h.cpu().numpy()          # the very taps the device was given
            c = e.add_coeff_dev(h, L * N)
            e.add_filter(in_ch=[i], out_ch=[o], coeff=c)
    torch.cuda.synchronize()
    e.finalize()
    return e, spot


def test_spot_check_two_outputs_against_oracle(full_engine, hip):
    ge, spot = full_engine
    outs = sorted({o for o, _ in spot})
    oe = bo.Engine(L, N, 4, I, len(outs))
    for c, f in enumerate(bo.interleaved_formats(FMT, I)):
        oe.set_format(0, c, f)
    for c, f in enumerate(bo.interleaved_formats("FLOAT_LE", len(outs))):
        oe.set_format(1, c, f)
    oe.out_bytes = len(outs) * L * 4
    for k, o in enumerate(outs):
        for i in range(I):
            oe.add_filter(in_ch=[i], out_ch=[k], coeff=oe.add_coeff(spot[(o, i)]))
    blocks = cases.raw_blocks(1234, N + 4, L, I, FMT)
    got = []
    for b, blk in enumerate(blocks):
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        gy = np.frombuffer(g.tobytes(), np.float32).reshape(L, O)[:, outs]
        got.append(gy.copy())
        if b >= N:
            oy = np.frombuffer(o.tobytes(), np.float32).reshape(L, len(outs))
            err = cases.rel_rms(gy, oy)
            assert err <= 1e-5, (b, err)
    # ... and, with nothing from oracle/ involved: the same two outputs as the float64 linear
    # convolution of the whole stream (scipy / pocketfft), every block from the first
    from scipy.signal import fftconvolve
    x = np.concatenate(blocks).astype(np.float64) / 8388608.0
    got = np.concatenate(got).astype(np.float64)
    for k, o in enumerate(outs):
        want = np.zeros(len(x))
        for i in range(I):
            want += fftconvolve(x[:, i], spot[(o, i)].astype(np.float64))[:len(x)]
        assert cases.rel_rms(got[:, k], want) <= 1e-5, o
    ab = ge.algorithmic_bytes()
    assert ab["mac"] == 65536.0 * (131072 + 2048 + 64)        # C*(F*P + U*P + O)
    # this IS the headline configuration: the code path bench.py times is the one checked here --
    # wave FFT, stream-ordered coefficients, deferred output
    # (tests/test_gpu_modes.py re-runs this file with the side-stream schedule forced: ping-pong, 3)
    assert ge.uses_wave_fft and ge.uses_stream_layout
    assert ge.block_mode == (3 if os.environ.get("BFHIP_OVERLAP") == "1" else 2)
    # ... including the device-buffer entry point with its fused [K3 of t-1 | K1 of t] launch: three
    # more blocks through bfhip_engine_block_dev, outputs owed until the next call / the sync
    torch = _torch()
    dev = torch.device("cuda", 0)
    more = cases.raw_blocks(4321, 3, L, I, FMT)
    srcs = [torch.from_numpy(b).to(dev) for b in more]
    outs_dev = [torch.zeros(L, O, dtype=torch.float32, device=dev) for _ in more]
    torch.cuda.synchronize()
    for k in range(len(more)):
        ge.block_dev(srcs[k], outs_dev[k])
    assert ge.sync() == 0
    for k, blk in enumerate(more):
        _, o = oe.block(blk)
        gy = outs_dev[k].cpu().numpy()[:, outs]
        oy = np.frombuffer(o.tobytes(), np.float32).reshape(L, len(outs))
        assert cases.rel_rms(gy, oy) <= 1e-5, k


def test_linearity_at_full_size(full_engine):
    """same engine state, three consecutive runs would mix histories; instead use the scaling
    law on ONE stream: an input scaled by 1/2 on the fly (exact in binary) must give outputs
    scaled by 1/2 from the moment the ring only holds scaled blocks"""
    ge, _ = full_engine
    blocks = cases.raw_blocks(77, 3, L, I, FMT, amplitude=0.05)
    blocks = [(b // 4) * 4 for b in blocks]                   # multiples of 4: halving is exact
    ref = []
    for k in range(N + 3):
        _, g = ge.block(blocks[k % 3])
        ref.append(np.frombuffer(g.tobytes(), np.float32).copy())
    half = []
    for k in range(N + 3):
        _, g = ge.block(blocks[k % 3] // 2)
        half.append(np.frombuffer(g.tobytes(), np.float32).copy())
    # after N blocks both rings hold the same periodic sequence (period 3 divides nothing of
    # N = 32, so compare equal phases): block N+j of each run has identical history up to x0.5
    for j in range(3):
        a, b = ref[N + j], half[N + j]
        assert np.array_equal(a * np.float32(0.5), b), j      # power-of-two scaling is bit-exact


def test_identity_crossbar_roundtrip_24bit(hip):
    """64 dirac filters on the diagonal, S24_4LE in and out: output == input up to the float32
    round trip through a 16384-point FFT pair (the reference itself is off by one LSB on ~1 %
    of 24-bit samples in this set-up, SURVEY B.5 i)"""
    e = hip.Engine(L, N, 4, I, O)
    e.set_interleaved(0, FMT)
    e.set_interleaved(1, FMT)
    for c in range(I):
        e.add_filter(in_ch=[c], out_ch=[c], coeff=-1)
    e.finalize()
    for blk in cases.raw_blocks(5, 3, L, I, FMT, amplitude=0.2):
        st, g = e.block(blk)
        assert st == 0
        d = np.abs(g.view(np.int32).reshape(L, O).astype(np.int64) - blk)
        assert d.max() <= 2 and (d > 0).mean() < 0.2, (d.max(), (d > 0).mean())
    for c in range(O):
        of = e.overflow(c)
        assert of.n_overflows == 0 and of.intlargest > 0


def test_config_d_shape_256_one_to_one_filters(hip):
    """BASELINE configs[3] shape on one GPU (massive_config style): 256 one-to-one filters of
    131072 taps (8192 x 16); 6 of the 256 channels recomputed by the oracle"""
    Ld, Nd, C = 8192, 16, 256
    ge = hip.Engine(Ld, Nd, 4, C, C)
    ge.set_interleaved(0, FMT)
    ge.set_interleaved(1, "FLOAT_LE")
    spot = [0, 7, 100, 101, 254, 255]
    irs = {}
    for c in range(C):
        h = cases.make_ir(np.random.default_rng(9000 + c), Ld * Nd, 1).astype(np.float32)
        if c in spot:
            irs[c] = h
        ge.add_filter(in_ch=[c], out_ch=[c], coeff=ge.add_coeff(h))
    ge.finalize()
    oe = bo.Engine(Ld, Nd, 4, len(spot), len(spot))
    oe.set_interleaved(0, FMT)
    oe.set_interleaved(1, "FLOAT_LE")
    for k, c in enumerate(spot):
        oe.add_filter(in_ch=[k], out_ch=[k], coeff=oe.add_coeff(irs[c]))
    for b, blk in enumerate(cases.raw_blocks(31, Nd + 3, Ld, C, FMT)):
        gs, g = ge.block(blk)
        os_, o = oe.block(np.ascontiguousarray(blk[:, spot]))
        assert gs == os_ == 0
        if b >= Nd:
            gy = np.frombuffer(g.tobytes(), np.float32).reshape(Ld, C)[:, spot]
            oy = np.frombuffer(o.tobytes(), np.float32).reshape(Ld, len(spot))
            assert cases.rel_rms(gy, oy) <= 1e-5, b


def test_config_e_shape_float64_million_taps(hip):
    """BASELINE configs[4] as uniform partitions (the reference only has uniform ones,
    SURVEY 0.2): 2-in/2-out, 1048576 taps = 8192 x 128, float64, <= 1e-12 relative RMS"""
    Le, Ne = 8192, 128
    spec = dict(L=Le, N=Ne, rs=8, n_in=2, n_out=2, infmt="FLOAT64_LE", outfmt="FLOAT64_LE",
                coeffs=[(cases.make_ir(np.random.default_rng(700 + k), Le * Ne, 2), 1.0, 0) for k in range(4)],
                filters=[dict(in_ch=[i], out_ch=[o], coeff=o * 2 + i) for o in range(2) for i in range(2)])
    ge, oe = cases.build(hip.Engine, spec), cases.build(bo.Engine, spec)
    blocks = cases.raw_blocks(8, 4, Le, 2, "FLOAT64_LE")
    for b in range(Ne + 3):
        blk = blocks[b % 4]
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        if b in (0, 1, Ne - 1, Ne, Ne + 1, Ne + 2):
            err = cases.rel_rms(np.frombuffer(g.tobytes(), np.float64), np.frombuffer(o.tobytes(), np.float64))
            assert err <= 1e-12, (b, err)


def test_every_coefficient_set_of_the_headline_configuration(full_engine):
    """All 64 outputs x all 64 inputs = all 4096 coefficient sets (8 GiB, the stream-ordered copy's
    every (entry, tile, partition) tile included): N silent blocks empty the rings, ONE block of
    independent noise on every input, N silent blocks; every output over those N + 1 blocks is the
    sum over the inputs of the linear convolution of that block with the impulse response loaded
    for the (output, input) pair -- computed in float64 with torch.fft on the device from the
    seeds the taps were generated from (nothing from oracle/ involved).  A set that sits at the
    wrong place contributes a different filter: every one of them is behind this number.  And the
    check itself is checked: one interior pair's filter replaced in the EXPECTATION must show."""
    torch = _torch()
    dev = torch.device("cuda", 0)
    ge, _ = full_engine
    silence = np.zeros((L, I), np.int32)
    noise = cases.raw_blocks(99, 1, L, I, FMT)[0]
    for _ in range(N):
        assert ge.block(silence)[0] == 0
    got = []
    for k in range(N + 1):
        st, g = ge.block(noise if k == 0 else silence)
        assert st == 0
        got.append(np.frombuffer(g.tobytes(), np.float32).reshape(L, O).copy())
    got = torch.from_numpy(np.concatenate(got)).to(dev).to(torch.float64)      # [(N+1) L][O]
    n_y, taps = (N + 1) * L, L * N
    n_fft = 1
    while n_fft < L + taps:
        n_fft *= 2
    x = torch.from_numpy(noise.astype(np.float64) / 8388608.0).to(dev)
    X = torch.fft.rfft(x, n=n_fft, dim=0)
    worst = 0.0
    for o in range(O):
        W = torch.zeros(n_fft // 2 + 1, dtype=torch.complex128, device=dev)
        for i in range(I):
            W += X[:, i] * torch.fft.rfft(_ir_dev(torch, 4321 + o * I + i, dev).to(torch.float64), n=n_fft)
        want = torch.fft.irfft(W, n=n_fft)[:n_y]
        err = float(torch.sqrt(((got[:, o] - want) ** 2).sum() / (want ** 2).sum()))
        worst = max(worst, err)
        assert err <= 1e-5, (o, err)
        if o == 37:
            # the expectation with ONE interior pair (37, 21) taking its neighbour's filter: the bar notices
            W2 = W + X[:, 21] * (torch.fft.rfft(_ir_dev(torch, 4321 + o * I + 22, dev).to(torch.float64), n=n_fft)
                                 - torch.fft.rfft(_ir_dev(torch, 4321 + o * I + 21, dev).to(torch.float64), n=n_fft))
            bad = torch.fft.irfft(W2, n=n_fft)[:n_y]
            assert float(torch.sqrt(((got[:, o] - bad) ** 2).sum() / (bad ** 2).sum())) > 1e-2
    assert worst > 0
