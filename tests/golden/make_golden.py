#!/usr/bin/env python3
"""Generate the committed golden vectors from the REFERENCE'S OWN CODE.

Runs only where /root/reference exists (this container): it calls the reference's inner
loops compiled into oracle/_ref/libbfref.so (recipe: oracle/Makefile, harness:
oracle/ref_harness.c) on seeded inputs and stores inputs + the reference's outputs as small
.npz fixtures.  The fixtures are data only; tests/test_oracle_golden.py replays them against
the oracle (CPU) and tests/test_gpu_ops.py against the HIP ops.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bforacle as bo  # noqa: E402

L = 64


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def ops(R, rs, seed):
    dt = np.float32 if rs == 4 else np.float64
    rng = np.random.default_rng(seed)
    R.ref_set_length(L, 0.0)
    out = {"L": np.int32(L)}
    # mixnscale, both modes, 1..6 buffers
    for n in (1, 2, 3, 4, 6):
        bufs = rng.standard_normal((n, 2 * L)).astype(dt)
        sc = rng.standard_normal(n)
        out["mix%d_in" % n] = bufs
        out["mix%d_scales" % n] = sc
        for mode, tag in ((1, "input"), (3, "output")):
            arr = (C.c_void_p * n)(*[bufs[i].ctypes.data for i in range(n)])
            o = np.empty(2 * L, dt)
            R.ref_mixnscale(rs, arr, p(o), (C.c_double * n)(*sc), n, mode)
            out["mix%d_%s" % (n, tag)] = o
    b = rng.standard_normal(2 * L).astype(dt)
    h = rng.standard_normal(2 * L).astype(dt)
    d = rng.standard_normal(2 * L).astype(dt)
    out.update(b=b, h=h, d=d)
    o = np.empty(2 * L, dt)
    R.ref_convolve(rs, p(b), p(h), p(o))
    out["convolve"] = o
    o = b.copy()
    R.ref_convolve_inplace(rs, p(o), p(h))
    out["convolve_inplace"] = o
    o = d.copy()
    R.ref_convolve_add(rs, p(b), p(h), p(o))
    out["convolve_add"] = o
    o = d.copy()
    R.ref_convolve_add_simd(rs, p(b), p(h), p(o))
    out["convolve_add_simd"] = o
    o = np.empty(2 * L, dt)
    R.ref_dirac_convolve(rs, p(b), p(o))
    out["dirac_convolve"] = o
    return out


RAWFMT = [  # name, bytes, sbytes, isfloat, swap
    ("S8", 1, 1, 0, 0), ("S16_LE", 2, 2, 0, 0), ("S16_BE", 2, 2, 0, 1),
    ("S24_LE", 3, 3, 0, 0), ("S24_BE", 3, 3, 0, 1), ("S24_4LE", 4, 3, 0, 0),
    ("S24_4BE", 4, 3, 0, 1), ("S32_LE", 4, 4, 0, 0), ("S32_BE", 4, 4, 0, 1),
    ("FLOAT_LE", 4, 4, 1, 0), ("FLOAT_BE", 4, 4, 1, 1),
    ("FLOAT64_LE", 8, 8, 1, 0), ("FLOAT64_BE", 8, 8, 1, 1),
]


def conversions(R, rs, seed):
    """raw2real and real2raw (no dither) for every sample format, spacing 3 (interleaved)"""
    dt = np.float32 if rs == 4 else np.float64
    rng = np.random.default_rng(seed)
    R.ref_set_length(L, 0.0)
    out = {}
    spacing = 3
    for name, nbytes, sbytes, isfloat, swap in RAWFMT:
        raw = rng.integers(0, 256, L * spacing * nbytes, dtype=np.uint8)
        if isfloat:
            # random bytes make NaNs; build finite floats instead
            vals = (rng.standard_normal(L * spacing) * 0.5).astype(np.float32 if nbytes == 4 else np.float64)
            raw = vals.view(np.uint8).copy()
            if swap:
                raw = raw.reshape(-1, nbytes)[:, ::-1].copy().ravel()
        real = np.empty(L, dt)
        R.ref_raw2real(rs, p(real), p(raw), nbytes, isfloat, spacing, swap, L)
        out["r2r_%s_raw" % name] = raw
        out["r2r_%s_real" % name] = real
        # real -> raw: samples spread so that some clip; int formats in output units
        full = float(1 << (8 * sbytes - 1)) if not isfloat else 1.0
        x = (rng.standard_normal(L) * 0.6 * full).astype(dt)
        x[:4] = np.array([0.0, -0.5, 0.5, -2.0], dt)            # exact-boundary cases
        of = bo.Overflow(0, 0, 0.0, 1.0 if isfloat else full - 1)
        o = np.zeros(L * spacing * nbytes, np.uint8)
        R.ref_real2raw(rs, p(o), p(x), sbytes * 8, nbytes, isfloat, spacing, swap, L, C.byref(of), -1)
        assert R.ref_exit_status() == 0
        out["rr_%s_x" % name] = x
        out["rr_%s_raw" % name] = o
        out["rr_%s_of" % name] = np.array(of.astuple(), np.float64)
    return out


def dither(R, rs, seed):
    """HP-TPDF dithered quantisation to S16, three channels, several consecutive blocks so
    that the error feedback state and the table walk (incl. a wrap) are exercised"""
    dt = np.float32 if rs == 4 else np.float64
    rng = np.random.default_rng(seed)
    R.ref_set_length(L, 0.0)
    n_ch, rate, nblk = 3, 100, 40          # small rate -> small table -> wraps quickly
    ok = R.ref_dither_init(n_ch, rate, rs, 0, L)
    assert ok
    tab = C.POINTER(C.c_int8)()
    n = R.ref_dither_table(C.byref(tab))
    table = np.ctypeslib.as_array(tab, (n,)).copy()
    xs = (rng.standard_normal((nblk, n_ch, L)) * 3000.0).astype(dt)
    xs[3, 1, 5] = 40000.0          # clips
    xs[7, 2, 9] = -50000.0
    raws = np.zeros((nblk, n_ch, L), np.int16)
    ofs = np.zeros((n_ch, 4))
    ptrs = np.zeros((nblk, n_ch), np.int32)
    of = [bo.Overflow(0, 0, 0.0, 32767.0) for _ in range(n_ch)]
    for b in range(nblk):
        for c in range(n_ch):
            o = np.zeros(L, np.int16)
            R.ref_real2raw(rs, p(o), p(xs[b, c]), 16, 2, 0, 1, 0, L, C.byref(of[c]), c)
            raws[b, c] = o
            ptrs[b, c] = R.ref_dither_randtab_ptr(c)
    for c in range(n_ch):
        ofs[c] = of[c].astuple()
    return {"table_head": table[:4096], "table_size": np.int32(n), "x": xs, "raw": raws,
            "of": ofs, "randtab_ptr": ptrs, "rate": np.int32(rate)}



def delay_and_window(R):
    """integer delay sequences (tests/test_oracle_delay.py CASES) and Kaiser windows through the
    reference's delay.c / firwindow.c"""
    sys.path.insert(0, os.path.dirname(HERE))
    import test_oracle_delay as tod
    out = {}
    for idx, case in enumerate(tod.CASES):
        F, ss, init, maxd, delays = case
        data = tod._data(case, idx)
        out["in%d" % idx] = data
        out["out%d" % idx] = tod._ref_run(R, F, ss, init, maxd, delays, data)
    for rs in (4, 8):
        dt = np.float32 if rs == 4 else np.float64
        for k, (ln, off) in enumerate([(63, 0.0), (63, 0.37), (63, -0.25), (64, 0.0), (31, 0.99), (31, -0.01)]):
            t = np.ones(ln, dt)
            R.ref_firwindow_kaiser(p(t), ln, off, 9.0, rs)
            out["kaiser%d_%d" % (rs, k)] = t
    np.savez_compressed(os.path.join(HERE, "ref_delay.npz"), **out)



def main():
    R = bo.ref()
    if R is None:
        raise SystemExit("oracle/_ref/libbfref.so is not built (needs /root/reference)")
    for rs, tag in ((4, "f32"), (8, "f64")):
        np.savez_compressed(os.path.join(HERE, "ref_ops_%s.npz" % tag), **ops(R, rs, 100 + rs))
        np.savez_compressed(os.path.join(HERE, "ref_conv_%s.npz" % tag), **conversions(R, rs, 200 + rs))
    if len(sys.argv) == 1:
        delay_and_window(bo.ref_delay())
        xtc_taps()
    # dither state is process-global in the reference: one precision per process
    which = sys.argv[1] if len(sys.argv) > 1 else None
    if which in ("f32", "f64"):
        rs = 4 if which == "f32" else 8
        np.savez_compressed(os.path.join(HERE, "ref_dither_%s.npz" % which), **dither(R, rs, 300 + rs))
    else:
        import subprocess
        for w in ("f32", "f64"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), w])
    print("golden vectors written to", HERE)



def xtc_taps():
    """The two coefficient files xtc_config names (`filename: "directpath.txt"` /
    "crosspath.txt", 4096 text taps each): DATA the reference ships, stored as a fixture so that
    the GPU box (no /root/reference) can run the reference's own cross-talk canceller."""
    ref = "/root/reference"
    np.savez_compressed(os.path.join(HERE, "xtc_taps.npz"),
                        directpath=np.loadtxt(os.path.join(ref, "directpath.txt")),
                        crosspath=np.loadtxt(os.path.join(ref, "crosspath.txt")))


if __name__ == "__main__":
    main()
