"""Shared builders: the same configuration is instantiated in the oracle (oracle/bforacle.py)
and in the HIP engine (brutefir_amd.Engine) -- both expose the same methods -- and driven with
the same seeded raw input blocks."""
import numpy as np

RAW_NP = {"S8": np.int8, "S16_LE": np.int16, "S32_LE": np.int32, "S24_4LE": np.int32,
          "FLOAT_LE": np.float32, "FLOAT64_LE": np.float64}


def make_ir(rng, taps, n_in):
    """white noise * exponential decay, normalised so that n_in summed filters stay inside
    +-1 (synthetic inputs)"""
    h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / (taps / 6.0))
    return h / (np.abs(h).sum() * n_in)


def crossbar(engine_cls, L, N, rs, I, O, infmt="S24_4LE", outfmt="S24_4LE", seed=4321,
             taps=None, delays=None, in_scale=None, out_scale=None, **kw):
    """I x O full crossbar with independent IRs (configs B and C are this shape)."""
    e = engine_cls(L, N, rs, I, O, **kw)
    e.set_interleaved(0, infmt)
    e.set_interleaved(1, outfmt)
    taps = L * N if taps is None else taps
    irs = {}
    for o in range(O):
        for i in range(I):
            rng = np.random.default_rng(seed + o * I + i)
            h = make_ir(rng, taps, I)
            irs[(o, i)] = h
            c = e.add_coeff(h)
            e.add_filter(in_ch=[i], out_ch=[o], coeff=c,
                         in_scale=[1.0 if in_scale is None else in_scale[o][i]],
                         out_scale=[1.0 if out_scale is None else out_scale[o][i]],
                         delayblocks=0 if delays is None else delays[o][i])
    if hasattr(e, "finalize"):
        e.finalize()
    return e, irs


def raw_blocks(seed, n_blocks, L, n_ch, fmt, amplitude=0.1):
    """seeded noise, frames x channels interleaved, in the raw sample format"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n_blocks * L, n_ch)) * amplitude
    if fmt.startswith("FLOAT"):
        raw = x.astype(RAW_NP[fmt])
    else:
        bits = {"S8": 8, "S16_LE": 16, "S24_4LE": 24, "S32_LE": 32}[fmt]
        raw = np.clip(np.round(x * (1 << (bits - 1))), -(1 << (bits - 1)), (1 << (bits - 1)) - 1)
        raw = raw.astype(RAW_NP[fmt])
    return [np.ascontiguousarray(raw[b * L:(b + 1) * L]) for b in range(n_blocks)]


def run(engine, blocks):
    outs, sts = [], []
    for b in blocks:
        st, raw = engine.block(b)
        sts.append(st)
        outs.append(raw)
    return sts, outs


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300)))


# ------------------------------------------------------------------ generic network builder

def build(engine_cls, spec, **kw):
    """spec: dict(L, N, rs, n_in, n_out, infmt, outfmt, coeffs=[(taps, scale, n_blocks)],
    filters=[dict(in_ch, in_scale, in_f, in_fscale, out_ch, out_scale, coeff, delayblocks,
    crossfade)], safety_limit) -> engine (oracle or HIP: same calls)"""
    e = engine_cls(spec["L"], spec["N"], spec["rs"], spec["n_in"], spec["n_out"], **kw)
    e.set_interleaved(0, spec["infmt"])
    e.set_interleaved(1, spec["outfmt"])
    if spec.get("safety_limit"):
        e.set_safety_limit(spec["safety_limit"])
    for taps, scale, nb in spec.get("coeffs", []):
        e.add_coeff(taps, scale, nb)
    for f in spec["filters"]:
        e.add_filter(**f)
    if hasattr(e, "finalize"):
        e.finalize()
    return e


def samples(raw, fmt):
    return np.frombuffer(raw.tobytes(), RAW_NP[fmt]).astype(np.float64)


def fade_cascade_network(engine_cls, rs):
    """One network that exercises delay bookkeeping, cascade evaluation and a cross-fade at once,
    and its float64 numpy model (scipy fftconvolve; nothing from oracle/):
       A: in0 -> ha, delayblocks 1            (cascade source)
       B: in1 -> hb                           (cascade source)
       C: 0.25*in0 + 0.5*A - 1.0*B -> hc, cross-fades to hd at block `sw` -> out0 (x 0.8)
       D: in1 -> hd, delayblocks 2 -> out1 and (x -0.5) out0
    Model: delays are whole blocks in front of the convolution (bfrun.c:1600) and cut the filter
    to the N - delay partitions that still fit the ring (cblocks, bfrun.c:1585-1591); a cascade
    adds no delay (convolve_eval, fftw_convolver.c:411-433); in the switch block the output is
    the ramp (1 - n/(L-1)) * old + n/(L-1) * new and afterwards the new taps act on all history
    (fftw_convolver.c:330-368, bfrun.c:1803-1838).  Returns (engine output, model, L, sw)."""
    from scipy.signal import fftconvolve
    dt = np.float32 if rs == 4 else np.float64
    L, N = 1024, 4
    rng = np.random.default_rng(21)
    ha, hb, hc, hd = [make_ir(rng, L * N - 17 * k, 2).astype(dt) for k in range(4)]
    e = engine_cls(L, N, rs, 2, 2)
    e.set_interleaved(0, "FLOAT64_LE")
    e.set_interleaved(1, "FLOAT_LE" if rs == 4 else "FLOAT64_LE")
    ca, cb, cc, cd = [e.add_coeff(h) for h in (ha, hb, hc, hd)]
    fa = e.add_filter(in_ch=[0], coeff=ca, delayblocks=1)
    fb = e.add_filter(in_ch=[1], coeff=cb)
    fc = e.add_filter(in_ch=[0], in_scale=[0.25], in_f=[fa, fb], in_fscale=[0.5, -1.0],
                      out_ch=[0], out_scale=[0.8], coeff=cc, crossfade=True)
    e.add_filter(in_ch=[1], out_ch=[1, 0], out_scale=[1.0, -0.5], coeff=cd, delayblocks=2)
    if hasattr(e, "finalize"):
        e.finalize()
    nblk, sw = 3 * N + 2, 2 * N + 1
    x = rng.standard_normal((nblk * L, 2)) * 0.1
    n = len(x)
    got = []
    for b in range(nblk):
        if b == sw:
            e.set_coeff(fc, cd)
        st, raw = e.block(x[b * L:(b + 1) * L])
        assert st == 0
        got.append(np.frombuffer(raw.tobytes(), dt).reshape(L, 2))
    y = np.concatenate(got).astype(np.float64)

    def conv(sig, h):
        return fftconvolve(np.asarray(sig, np.float64), np.asarray(h, np.float64))[:n]

    def delayed(sig, blocks):
        return np.concatenate([np.zeros(blocks * L), sig[:n - blocks * L]])
    ya = conv(delayed(x[:, 0], 1), ha[:(N - 1) * L])
    yb = conv(x[:, 1], hb)
    in_c = 0.25 * x[:, 0] + 0.5 * ya - yb
    yc_old, yc_new = conv(in_c, hc), conv(in_c, hd)
    w = np.arange(L) / (L - 1.0)
    yc = yc_old.copy()
    s = slice(sw * L, (sw + 1) * L)
    yc[s] = yc_old[s] * (1 - w) + yc_new[s] * w
    yc[(sw + 1) * L:] = yc_new[(sw + 1) * L:]
    yd = conv(delayed(x[:, 1], 2), hd[:(N - 2) * L])
    return y, np.stack([0.8 * yc - 0.5 * yd, yd], axis=1), L, sw
