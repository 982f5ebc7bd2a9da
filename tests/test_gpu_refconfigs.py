"""GPU: the configurations the reference ships (bench1_config ... bench5_config, massive_config,
xtc_config, and the stock `filter_length: 65536`), rebuilt through the C ABI exactly as bfconf
would hand them to the filter process -- channel layouts of the opened devices
(`channels: 26/24,25`, packed S24_LE, several devices in one block buffer), N:1 input mappings,
per-input and per-output scales in dB / linear, `delay:`, `blocks:`, dither flags, cross-fade
flags and the cli script of bench5 -- and run against the oracle.  The coefficient files
("dirac pulse", the xtc text files) are replaced by what they contain or by seeded responses;
the inputs are seeded noise instead of /dev/zero."""
import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu
SB = {"S8": 1, "S16_LE": 2, "S24_LE": 3, "S24_4LE": 4, "S32_LE": 4, "FLOAT_LE": 4, "FLOAT64_LE": 8}


def db(x):                       # `/<dB>` (bfconf.c:931-951): attenuation in dB
    return 10.0 ** (-x / 20.0)


class Config:
    """what bfconf + dai leave behind for the filter process"""

    def __init__(self, L, N, rs=4):
        self.L, self.N, self.rs = L, N, rs
        self.fmt = [[], []]                # physical channels: (format name, spacing, byte offset)
        self.bytes = [0, 0]
        self.v2p = [None, None]
        self.coeffs, self.filters, self.dither, self.rate = [], [], [], 44100

    def device(self, io, name, open_channels, used):
        """one input/output device section: `channels: open/used...`, interleaved frames, placed
        behind the devices declared before it in the block buffer (dai.c:537-576)"""
        base = self.bytes[io]
        for u in used:
            self.fmt[io].append((name, open_channels, base + u * SB[name]))
        self.bytes[io] = base + open_channels * SB[name] * self.L

    def build(self, mod):
        n_virt = [len(self.v2p[io]) if self.v2p[io] else len(self.fmt[io]) for io in range(2)]
        e = mod.Engine(self.L, self.N, self.rs, n_virt[0], n_virt[1])
        for io in range(2):
            if self.v2p[io]:
                e.map_channels(io, self.v2p[io])
            for c, (name, spacing, off) in enumerate(self.fmt[io]):
                e.set_format(io, c, mod.make_format(name, spacing, off))
        e.in_bytes, e.out_bytes = self.bytes
        for taps, scale, nb in self.coeffs:
            e.add_coeff(np.asarray(taps, np.float32 if self.rs == 4 else np.float64), scale, nb)
        if self.dither:
            e.enable_dither(self.dither, self.rate, 0)
        for f in self.filters:
            e.add_filter(**f)
        if hasattr(e, "finalize"):
            e.finalize()
        return e


def _noise_inputs(seed, cfg, n_blocks):
    """per block: one raw buffer; every physical input channel gets its own seeded noise"""
    rng = np.random.default_rng(seed)
    blocks = []
    for _ in range(n_blocks):
        buf = np.zeros(cfg.bytes[0], np.uint8)
        for name, spacing, off in cfg.fmt[0]:
            nb = SB[name]
            x = rng.standard_normal(cfg.L) * 0.1
            if name.startswith("FLOAT"):
                raw = x.astype(np.float32 if nb == 4 else np.float64).view(np.uint8).reshape(cfg.L, nb)
            else:
                bits = {"S8": 8, "S16_LE": 16, "S24_LE": 24, "S24_4LE": 24, "S32_LE": 32}[name]
                q = np.clip(np.round(x * (1 << (bits - 1))), -(1 << (bits - 1)), (1 << (bits - 1)) - 1).astype(np.int64)
                raw = np.stack([(q >> (8 * k)) & 0xff for k in range(nb)], axis=1).astype(np.uint8)
                if name == "S24_4LE":
                    raw[:, 3] = np.where(q < 0, 0xff, 0).astype(np.uint8)      # sign extension byte
            idx = off + np.arange(cfg.L)[:, None] * (spacing * nb) + np.arange(nb)[None, :]
            buf[idx] = raw
        blocks.append(buf)
    return blocks


def _decode(buf, cfg, io):
    out = []
    for name, spacing, off in cfg.fmt[io]:
        nb = SB[name]
        idx = off + np.arange(cfg.L)[:, None] * (spacing * nb) + np.arange(nb)[None, :]
        raw = np.ascontiguousarray(buf[idx])
        if name.startswith("FLOAT"):
            out.append(raw.view(np.float32 if nb == 4 else np.float64).ravel().astype(np.float64))
        else:
            v = np.zeros(cfg.L, np.int64)
            for k in range(min(nb, 3 if name == "S24_4LE" else nb)):
                v |= raw[:, k].astype(np.int64) << (8 * k)
            bits = {"S8": 8, "S16_LE": 16, "S24_LE": 24, "S24_4LE": 24, "S32_LE": 32}[name]
            v = np.where(v >= (1 << (bits - 1)), v - (1 << bits), v)
            out.append(v.astype(np.float64))
    return out


def _run(hip, cfg, n_blocks, seed=0, script=None, lsb_tol=2.0, float_tol=1e-5):
    # 24-bit outputs of a float32 engine sit at its rounding noise: either implementation is within
    # 1 LSB of the exact result per FFT round trip (float32 has 24 bits, a round trip costs ~1e-6
    # relative), so two of them may differ by 2 (4 behind a cascade, or with the error-feedback
    # dither quantiser, where a single flip moves the following samples too).  Plain 16-bit
    # outputs agree within 1, dithered ones within 2 for the same reason.
    ge, oe = cfg.build(hip), cfg.build(bo)
    for b, blk in enumerate(_noise_inputs(seed, cfg, n_blocks)):
        if script:
            script(b, ge)
            script(b, oe)
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0, b
        for ch, (gv, ov) in enumerate(zip(_decode(g, cfg, 1), _decode(o, cfg, 1))):
            if cfg.fmt[1][ch][0].startswith("FLOAT"):
                lvl = max(float(np.sqrt((ov ** 2).mean())), 1e-3)
                assert float(np.sqrt(((gv - ov) ** 2).mean())) <= float_tol * lvl, (b, ch)
            else:
                assert np.abs(gv - ov).max() <= lsb_tol, (b, ch, np.abs(gv - ov).max())
    for ch in range(ge.n_out):
        a, c = ge.overflow(ch), oe.overflow(ch)
        assert a.n_overflows == c.n_overflows and a.max == c.max, ch
    return ge


DIRAC = [1.0]


def _one_to_one(cfg, n, crossfade=False):
    cfg.coeffs = [(DIRAC, 1.0, 0)]
    cfg.filters = [dict(in_ch=[i], out_ch=[i], coeff=0, crossfade=crossfade) for i in range(n)]


def test_bench1_config(hip):
    """two inputs -> four input filters -> two output filters (from_filters), dirac coefficients"""
    cfg = Config(8192, 8)
    cfg.device(0, "S24_4LE", 2, [0, 1])
    cfg.device(1, "S24_4LE", 2, [0, 1])
    cfg.coeffs = [(DIRAC, 1.0, 0)] * 6
    # the reference sorts filters topologically (bfconf.c:2933-2964): 2,3,4,5 then 0,1
    cfg.filters = [dict(in_ch=[0], coeff=2), dict(in_ch=[0], coeff=3), dict(in_ch=[1], coeff=4), dict(in_ch=[1], coeff=5),
                   dict(in_f=[0, 3], out_ch=[0], coeff=0), dict(in_f=[1, 2], out_ch=[1], coeff=1)]
    _run(hip, cfg, 10, lsb_tol=4.0)


def test_bench2_config(hip):
    cfg = Config(8192, 8)
    cfg.device(0, "S24_4LE", 26, range(26))
    cfg.device(1, "S24_4LE", 26, range(26))
    _one_to_one(cfg, 26)
    _run(hip, cfg, 4)


def test_bench3_config_stock_filter_length(hip):
    """`filter_length: 65536;` -- one partition, also BruteFIR's stock default (bfconf.c:197)"""
    cfg = Config(65536, 1)
    cfg.device(0, "S24_4LE", 26, range(26))
    cfg.device(1, "S24_4LE", 26, range(26))
    _one_to_one(cfg, 26)
    _run(hip, cfg, 3)


def test_bench4_config(hip):
    """4096 x 13; six virtual inputs on two channels of a four-channel S24_4LE device
    (`channels: 4/0,3; mapping: 0,1,0,1,0,1`) plus an S8 device; S16_LE output with dither and a
    float output; dB and linear scales, `delay: 1`, a `blocks: 1` set, a three-level cascade"""
    cfg = Config(4096, 13)
    cfg.device(0, "S24_4LE", 4, [0, 3])
    cfg.device(0, "S8", 2, [0, 1])
    cfg.v2p[0] = [0, 1, 0, 1, 0, 1, 2, 3]
    cfg.device(1, "S16_LE", 1, [0])
    cfg.device(1, "FLOAT_LE", 1, [0])
    cfg.dither = [0]
    cfg.coeffs = [(DIRAC, 1.0, 0), (DIRAC, 1.0, 1)]
    cfg.filters = [
        dict(in_ch=[0, 1], in_scale=[-1.0, 2.0], delayblocks=1, out_ch=[0], out_scale=[db(3)], coeff=0),
        dict(in_ch=[6, 2], in_f=[0], in_fscale=[db(3)], out_ch=[1], out_scale=[db(5.32)], coeff=1),
        dict(in_f=[0, 1], in_ch=[3, 4, 5, 7], out_ch=[0, 1], coeff=0),
    ]
    # (the S16 output clips here -- six inputs summed with gain -- and the dither quantiser feeds
    # its error back: float32 rounding differences in front of it show up as a few LSB)
    _run(hip, cfg, 2 * 13 + 3, lsb_tol=4.0, float_tol=3e-5)


def test_bench5_config_cli_script(hip):
    """bench2 with `crossfade: true` everywhere and the cli script that switches every filter
    between coefficient 0 and the dirac (-1) block after block"""
    cfg = Config(8192, 8)
    cfg.device(0, "S24_4LE", 26, range(26))
    cfg.device(1, "S24_4LE", 26, range(26))
    _one_to_one(cfg, 26, crossfade=True)

    def script(b, eng):
        for f in range(26):
            eng.set_coeff(f, 0 if b % 2 == 0 else -1)
    _run(hip, cfg, 6, script=script)


def test_massive_config(hip):
    """26 channels of packed S24_LE with dither: true, 8192 x 16"""
    cfg = Config(8192, 16)
    cfg.device(0, "S24_LE", 26, range(26))
    cfg.device(1, "S24_LE", 26, range(26))
    cfg.dither = list(range(26))
    _one_to_one(cfg, 26)
    _run(hip, cfg, 3, lsb_tol=4.0)


def test_xtc_config(hip):
    """64 x 64 cross-talk canceller: channels 24 and 25 of a 26-channel packed S24_LE device on
    both sides (`channels: 26/24,25`), two coefficient sets, dither: true"""
    cfg = Config(64, 64)
    cfg.device(0, "S24_LE", 26, [24, 25])
    cfg.device(1, "S24_LE", 26, [24, 25])
    cfg.dither = [0, 1]
    rng = np.random.default_rng(3)
    cfg.coeffs = [(cases.make_ir(rng, 4096, 2), 1.0, 0), (cases.make_ir(rng, 4096, 2) * 0.5, 1.0, 0)]
    cfg.filters = [dict(in_ch=[0], out_ch=[0], coeff=0), dict(in_ch=[1], out_ch=[0], coeff=1),
                   dict(in_ch=[1], out_ch=[1], coeff=0), dict(in_ch=[0], out_ch=[1], coeff=1)]
    ge = _run(hip, cfg, 2 * 64 + 5, lsb_tol=4.0)
    # what is not ours to write stays untouched: the other 24 channels of the output frames
    st, g = ge.block(_noise_inputs(99, cfg, 1)[0])
    frames = g.reshape(cfg.L, 26, 3)
    assert not frames[:, :24, :].any()
