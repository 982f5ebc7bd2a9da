"""GPU: the reference's OWN filter_process() -- bfrun.c compiled unchanged from /root/reference into
oracle/_ref/ref_filter_process (oracle/ref_filter_process_harness.c; built in the build container, the
binary travels) -- run as a forked filter process over the PRODUCT's 22 convolver.h symbols
(libbfhip.so, the unfused path: per-block ops on the GPU), against

  * the fused engine (bfhip_engine_block), and
  * the oracle's restatement of filter_process(),

on random filter networks with run-time control sequences (tests/test_gpu_fuzz.py's generator:
channel and filter inputs, cascades through convolve_eval, several outputs per filter, outputs fed
by several filters, negative / zero scales, coeff -1, short sets, delayblocks beyond N - 1,
cross-fading switches; fctrl writes between blocks).

What this pins, and did not have a reference OUTPUT to be pinned by before: the block-ring / delay /
cblocks / warm-up bookkeeping (`(blockcounter + delay) % n_blocks`, the clamp, the
`procblocks` guard), the coefficient switch with its one-block cross-fade (A7), the cascade (A8), the
input and output mixes as filter_process() drives them.  A wrong slot or a fade on the wrong block
is a gross error, not a rounding one.  What it does not pin: the FFT itself -- both sides of the
convolver.h boundary are this repository's transforms (FFTW is absent)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import bforacle as bo
import cases
import test_gpu_fuzz as fuzz

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_filter_process")
EXE_PATCHED = os.path.join(ROOT, "oracle", "_ref", "ref_filter_process_bfhip")
FMT_CODE = {"S16_LE": 0, "S24_4LE": 1, "S32_LE": 2, "FLOAT_LE": 3, "FLOAT64_LE": 4, "S24_LE": 5}
KIND = {"coeff": 0, "scale_in": 1, "scale_out": 2, "fscale": 3, "delay": 4}


UNDEF_SUBDELAY = -100          # BF_UNDEFINED_SUBDELAY (bfmod.h:89-90)


def _write_spec(path, spec, blocks, events, f_owner=None):
    """spec["channels"] (optional): dict(maps=[v2p_in, v2p_out], delay, maxdelay, mute, subdelay (each
    [per virtual input, per virtual output]), dither=[flag per physical output], sdf_length, sdf_beta)"""
    rs = spec["rs"]
    dt = np.float32 if rs == 4 else np.float64
    evs = []
    for b, lst in sorted(events.items()):
        for ev in lst:
            if ev[0] == "coeff":
                evs.append((b, 0, ev[1], 0, float(ev[2])))
            elif ev[0] == "scale":
                evs.append((b, 1 if ev[2] == 0 else 2, ev[1], ev[3], float(ev[4])))
            elif ev[0] == "fscale":
                evs.append((b, 3, ev[1], ev[2], float(ev[3])))
            elif ev[0] == "delayblocks" or (ev[0] == "delay" and len(ev) == 3):
                evs.append((b, 4, ev[1], 0, float(ev[2])))
            elif ev[0] == "mute":                       # ("mute", io, virtual channel, on)
                evs.append((b, 5, ev[1], ev[2], float(ev[3])))
            elif ev[0] == "delay":                      # ("delay", io, virtual channel, samples)
                evs.append((b, 6, ev[1], ev[2], float(ev[3])))
            elif ev[0] == "subdelay":                   # ("subdelay", io, virtual channel, slots)
                evs.append((b, 7, ev[1], ev[2], float(ev[3])))
            else:                                       # ("rewrite", coefficient set, partition, gain)
                evs.append((b, 8, ev[1], ev[2], float(ev[3])))
    with open(path, "wb") as f:
        n_procs = 1 + (max(f_owner) if f_owner else 0)
        ch = spec.get("channels")
        f.write(struct.pack("<13i", 0x42465253 if ch else 0x42465251, spec["L"], spec["N"], rs, spec["n_in"], spec["n_out"],
                            FMT_CODE[spec["infmt"]], FMT_CODE[spec["outfmt"]], len(spec["coeffs"]),
                            len(spec["filters"]), len(blocks), len(evs), n_procs))
        if ch:
            for io in range(2):
                f.write(struct.pack("<i", max(ch["maps"][io]) + 1))
                for key in ("maps", "delay", "maxdelay", "mute", "subdelay"):
                    f.write(np.asarray(ch[key][io], np.int32).tobytes())
            f.write(np.asarray(ch["dither"], np.int32).tobytes())
            f.write(struct.pack("<id", ch.get("sdf_length", -1), ch.get("sdf_beta", 9.0)))
        for taps, scale, nb in spec["coeffs"]:
            t = np.ascontiguousarray(taps, dt)
            f.write(struct.pack("<iid", len(t), nb, scale))
            f.write(t.tobytes())
        for fi, fl in enumerate(spec["filters"]):
            f.write(struct.pack("<7i", f_owner[fi] if f_owner else 0, len(fl["in_ch"]), len(fl["in_f"]), len(fl["out_ch"]),
                                fl["coeff"], fl["delayblocks"], int(fl["crossfade"])))
            f.write(np.asarray(fl["in_ch"], np.int32).tobytes())
            f.write(np.asarray(fl["in_scale"], np.float64).tobytes())
            f.write(np.asarray(fl["in_f"], np.int32).tobytes())
            f.write(np.asarray(fl["in_fscale"], np.float64).tobytes())
            f.write(np.asarray(fl["out_ch"], np.int32).tobytes())
            f.write(np.asarray(fl["out_scale"], np.float64).tobytes())
        for ev in evs:
            f.write(struct.pack("<4id", *ev))
        for blk in blocks:
            f.write(np.ascontiguousarray(blk).tobytes())


@pytest.mark.parametrize("seed", range(int(os.environ.get("BFHIP_REFLOOP_SEEDS", "24"))))
def test_reference_filter_process_over_the_product_vs_fused_engine_and_oracle(hip, tmp_path, seed):
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/ref_filter_process is missing: it is built from the reference's bfrun.c in the build "
                    "container (make -C oracle ref, __graft_entry__.build) and travels to the GPU box")
    spec, n_blocks, events = fuzz._network(seed + 7000)
    blocks = cases.raw_blocks(seed, n_blocks, spec["L"], spec["n_in"], spec["infmt"], amplitude=0.2)
    _write_spec(tmp_path / "spec.bin", spec, blocks, events)
    r = subprocess.run([EXE, str(tmp_path / "spec.bin"), str(tmp_path / "out.raw")], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    odt = np.float32 if spec["rs"] == 4 else np.float64
    ref = np.fromfile(tmp_path / "out.raw", odt).reshape(n_blocks, spec["L"], spec["n_out"]).astype(np.float64)
    ge, oe = cases.build(hip.Engine, spec), cases.build(bo.Engine, spec)
    tol = 3e-5 if spec["rs"] == 4 else 1e-11
    scale = 0.0
    for b, blk in enumerate(blocks):
        fuzz._apply(ge, events.get(b, []))
        fuzz._apply(oe, events.get(b, []))
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0, (seed, b)
        gsamp = cases.samples(g, spec["outfmt"]).reshape(spec["L"], spec["n_out"])
        osamp = cases.samples(o, spec["outfmt"]).reshape(spec["L"], spec["n_out"])
        scale = max(scale, float(np.abs(ref[b]).max()))
        for ch in range(spec["n_out"]):
            want = ref[b][:, ch]
            lvl = max(float(np.sqrt((want ** 2).mean())), 1e-3 * scale, fuzz.FLOOR)
            for who, got in (("fused engine", gsamp[:, ch]), ("oracle", osamp[:, ch])):
                err = float(np.sqrt(((got - want) ** 2).mean()))
                assert err <= tol * lvl, (who, seed, b, ch, err, lvl, spec["filters"], events)
    # the overflow structs the reference's loop kept in icomm against the engine's
    for ln in r.stdout.splitlines():
        if ln.startswith("output "):
            ch = int(ln.split()[1].rstrip(":"))
            assert int(ln.split()[2]) == ge.overflow(ch).n_overflows, ln


def _named_specs():
    """configurations the reference ships, within what the harness sets up (one interleaved device per
    side, no N:1 channels, no dither): bench1_config's cascade, bench5_config's cli script, and
    bench4_config's ring of 13 partitions with its delays -- plus a 16-bit output driven into clipping"""
    rng = np.random.default_rng(99)
    dirac = np.array([1.0], np.float32)
    out = {}
    # bench1_config: two inputs -> four input filters -> two output filters (from_filters), dirac coefficients
    out["bench1"] = (dict(L=8192, N=8, rs=4, n_in=2, n_out=2, infmt="S24_4LE", outfmt="S24_4LE",
                          coeffs=[(dirac, 1.0, 0)] * 6,
                          filters=[_f(in_ch=[0], coeff=2), _f(in_ch=[0], coeff=3), _f(in_ch=[1], coeff=4), _f(in_ch=[1], coeff=5),
                                   _f(in_f=[0, 3], out_ch=[0], coeff=0), _f(in_f=[1, 2], out_ch=[1], coeff=1)]), 10, {}, 4.0)
    # bench5_config: 26 cross-fading one-to-one filters toggled between coefficient 0 and the dirac (-1) every block
    ev = {b: [("coeff", f, 0 if b % 2 == 0 else -1) for f in range(26)] for b in range(1, 7)}
    out["bench5"] = (dict(L=8192, N=8, rs=4, n_in=26, n_out=26, infmt="S24_4LE", outfmt="S24_4LE",
                          coeffs=[(cases.make_ir(rng, 8192 * 3, 1), 1.0, 0)],
                          filters=[_f(in_ch=[i], out_ch=[i], coeff=0, crossfade=True) for i in range(26)]), 7, ev, 4.0)   # (a fade block: three more float32 round trips)
    # bench4_config's partitioning: 4096 x 13 (a ring that is no power of two), delays up to and beyond N - 1
    flt = [_f(in_ch=[i], out_ch=[o], coeff=(o + i) % 3, delayblocks=[0, 1, 5, 12, 40][(o * 3 + i) % 5],
              in_scale=[[1.0, -0.5, 0.25][(o + i) % 3]]) for o in range(4) for i in range(3)]
    ev = {14: [("delay", 2, 7), ("coeff", 5, 1)], 20: [("delay", 2, 0), ("scale", 7, 0, 0, 2.0)], 30: [("coeff", 5, -1)]}
    out["ring13"] = (dict(L=4096, N=13, rs=8, n_in=3, n_out=4, infmt="S24_4LE", outfmt="FLOAT64_LE",
                          coeffs=[(cases.make_ir(rng, 4096 * 13, 3), 1.0, 0), (cases.make_ir(rng, 4096 * 4 + 100, 3), 0.5, 0),
                                  (cases.make_ir(rng, 4096 * 13, 3), -1.0, 2)], filters=flt), 2 * 13 + 8, ev, None)
    # 16-bit output, loud: the clip and the overflow counters of icomm->overflow
    out["clip16"] = (dict(L=1024, N=4, rs=4, n_in=2, n_out=2, infmt="S24_4LE", outfmt="S16_LE",
                          coeffs=[(cases.make_ir(rng, 1024 * 4, 1) * 12.0, 1.0, 0)],
                          filters=[_f(in_ch=[i], out_ch=[o], coeff=0) for o in range(2) for i in range(2)]), 12, {}, 1.0)
    return out


def _f(in_ch=(), in_scale=None, in_f=(), in_fscale=None, out_ch=(), out_scale=None, coeff=-1, delayblocks=0, crossfade=False):
    return dict(in_ch=list(in_ch), in_scale=list(in_scale) if in_scale else [1.0] * len(in_ch), in_f=list(in_f),
                in_fscale=list(in_fscale) if in_fscale else [1.0] * len(in_f), out_ch=list(out_ch),
                out_scale=list(out_scale) if out_scale else [1.0] * len(out_ch), coeff=coeff, delayblocks=delayblocks, crossfade=crossfade)


@pytest.mark.parametrize("name", ["bench1", "bench5", "ring13", "clip16"])
def test_reference_filter_process_on_the_reference_s_own_configurations(hip, tmp_path, name):
    if not os.path.exists(EXE):
        pytest.fail("oracle/_ref/ref_filter_process is missing (built from the reference's bfrun.c in the build container)")
    spec, n_blocks, events, lsb_tol = _named_specs()[name]
    amp = 0.9 if name == "clip16" else 0.2
    blocks = cases.raw_blocks(5, n_blocks, spec["L"], spec["n_in"], spec["infmt"], amplitude=amp)
    _write_spec(tmp_path / "spec.bin", spec, blocks, events)
    r = subprocess.run([EXE, str(tmp_path / "spec.bin"), str(tmp_path / "out.raw")], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    odt = {"S24_4LE": np.int32, "S16_LE": np.int16, "FLOAT64_LE": np.float64}[spec["outfmt"]]
    ref = np.fromfile(tmp_path / "out.raw", odt).reshape(n_blocks, spec["L"], spec["n_out"])
    ge = cases.build(hip.Engine, spec)
    for b, blk in enumerate(blocks):
        fuzz._apply(ge, events.get(b, []))
        st, g = ge.block(blk)
        assert st == 0
        got = np.frombuffer(g.tobytes(), odt).reshape(spec["L"], spec["n_out"])
        if lsb_tol is None:
            assert cases.rel_rms(got, ref[b]) <= 1e-11 or np.abs(ref[b]).max() == 0, (name, b)
        else:
            assert np.abs(got.astype(np.int64) - ref[b].astype(np.int64)).max() <= lsb_tol, (name, b)
    counted = 0
    for ln in r.stdout.splitlines():
        if ln.startswith("output "):
            ch, n_over = int(ln.split()[1].rstrip(":")), int(ln.split()[2])
            mine = ge.overflow(ch).n_overflows
            counted += n_over
            # a sample within one count of the clip level may fall on either side of it
            assert abs(n_over - mine) <= max(2, n_over // 100), (name, ln, mine)
    assert counted > 0 or name != "clip16"


def _run_host(exe, tmp_path, tag, spec, blocks, events, f_owner=None, env=None):
    _write_spec(tmp_path / ("spec_%s.bin" % tag), spec, blocks, events, f_owner)
    out = tmp_path / ("out_%s.raw" % tag)
    r = subprocess.run([exe, str(tmp_path / ("spec_%s.bin" % tag)), str(out)], capture_output=True, text=True, timeout=90,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (tag, r.stdout[-500:], r.stderr[-1500:])
    return open(out, "rb").read(), [ln for ln in r.stdout.splitlines() if ln.startswith("output ")], r.stderr


@pytest.mark.parametrize("seed", range(int(os.environ.get("BFHIP_REFLOOP_SEED0", "0")),
                                        int(os.environ.get("BFHIP_REFLOOP_SEED0", "0")) + int(os.environ.get("BFHIP_REFLOOP_SEEDS", "24")) // 2))
def test_the_patched_filter_process_runs_and_its_processes_agree(hip, tmp_path, seed):
    """patches/bfrun-bfhip.diff applied to the reference's bfrun.c and RUN (oracle/_ref/ref_filter_process_bfhip,
    -DBF_HAVE_BFHIP): the patched filter_process() takes the fused path -- bfhip_setup() builds the engine
    from bfconf / icomm / dai_buffer_format in the forked filter process, bfhip_period() forwards the fctrl
    snapshot and runs the block.  Against the UNPATCHED filter_process() on the same network and control
    sequence (same tolerance as everywhere), and, with the filters dealt out over two and three filter
    processes the way bfconf does (connected filters and mixed outputs together, bfconf.c:2893-2931; all
    processes on the one GPU of the box): **byte-identical** to the patched one-process run -- the host
    patch's multi-process path, end to end, through the reference's own process code."""
    from test_gpu_shards import _assign
    for exe in (EXE, EXE_PATCHED):
        if not os.path.exists(exe):
            pytest.fail("%s is missing (built from the reference's bfrun.c in the build container)" % exe)
    spec, n_blocks, events = fuzz._network(seed + 9000)
    if seed % 3 == 0:
        spec["outfmt"] = "S24_4LE"
    blocks = cases.raw_blocks(seed, n_blocks, spec["L"], spec["n_in"], spec["infmt"], amplitude=0.2)
    plain, plain_of, _ = _run_host(EXE, tmp_path, "plain", spec, blocks, events)
    one, one_of, _ = _run_host(EXE_PATCHED, tmp_path, "p1", spec, blocks, events)
    dis, _, _ = _run_host(EXE_PATCHED, tmp_path, "disabled", spec, blocks, events, env={"BFHIP_DISABLE": "1"})
    assert dis == plain                          # BFHIP_DISABLE=1: the patched host takes the unfused path, bit for bit
    rng = np.random.default_rng(seed)
    for n_proc in (2, 3):
        f_owner, _ = _assign(spec, n_proc, rng)
        if len(set(f_owner)) != n_proc:
            continue                             # a process without filters: bfconf never creates one (bfconf.c:2227-2318)
        many, many_of, _ = _run_host(EXE_PATCHED, tmp_path, "p%d" % n_proc, spec, blocks, events, f_owner)
        assert many == one, (seed, n_proc, f_owner)
        assert many_of == one_of, (seed, n_proc)
        ref_many, _, _ = _run_host(EXE, tmp_path, "u%d" % n_proc, spec, blocks, events, f_owner)
        assert ref_many == plain, (seed, n_proc)         # the reference's own multi-process = single-process equality (SURVEY B.5 iii), over the product's ops
    odt = {"FLOAT_LE": np.float32, "FLOAT64_LE": np.float64, "S24_4LE": np.int32}[spec["outfmt"]]
    a = np.frombuffer(plain, odt).reshape(n_blocks, spec["L"], spec["n_out"]).astype(np.float64)
    b = np.frombuffer(one, odt).reshape(n_blocks, spec["L"], spec["n_out"]).astype(np.float64)
    if spec["outfmt"] == "S24_4LE":
        assert np.abs(a - b).max() <= 4.0, seed          # (24-bit output of a float32 engine: rounding noise of the round trips)
    else:
        tol = 3e-5 if spec["rs"] == 4 else 1e-11
        scale = float(np.abs(a).max())
        for blk in range(n_blocks):
            for ch in range(spec["n_out"]):
                lvl = max(float(np.sqrt((a[blk][:, ch] ** 2).mean())), 1e-3 * scale, fuzz.FLOOR)
                assert float(np.sqrt(((a[blk][:, ch] - b[blk][:, ch]) ** 2).mean())) <= tol * lvl, (seed, blk, ch)


def test_the_patched_host_prints_its_benchmark_table_with_device_times(hip, tmp_path):
    """`benchmark: true;` -- the reference prints one row per filter process and ten periods with the
    time of every stage (bfrun.c:2035-2078).  The patched filter_process() adds the engine's
    bfhip_engine_stage_times() into t[0..6] before that print: the table the user knows, printed by the
    reference's own code, with device milliseconds in it (stages fused away read 0)."""
    if not os.path.exists(EXE_PATCHED):
        pytest.fail("oracle/_ref/ref_filter_process_bfhip is missing")
    L, N, I, O = 4096, 4, 3, 4
    rng = np.random.default_rng(1)
    spec = dict(L=L, N=N, rs=4, n_in=I, n_out=O, infmt="S24_4LE", outfmt="S24_4LE",
                coeffs=[(cases.make_ir(rng, L * N, I), 1.0, 0)],
                filters=[_f(in_ch=[i], out_ch=[o], coeff=0) for o in range(O) for i in range(I)] +
                        [_f(in_ch=[0, 1], in_scale=[0.5, 0.5], out_ch=[O - 1], coeff=0)])
    blocks = cases.raw_blocks(2, 31, L, I, "S24_4LE")
    f_owner = [0 if f["out_ch"][0] < 2 else 1 for f in spec["filters"]]
    for tag, owner in (("one", None), ("two", f_owner)):
        _, _, err = _run_host(EXE_PATCHED, tmp_path, "bm_" + tag, spec, blocks, {}, owner, env={"BFREF_BENCHMARK": "1"})
        rows = [[c.strip() for c in ln.split("|")] for ln in err.splitlines() if ln.count("|") == 10 and "raw2real" not in ln]
        n_proc = 2 if owner else 1
        assert len(rows) == n_proc * 3, err[-1500:]                  # 31 periods: three prints per process
        assert len({row[0] for row in rows}) == n_proc
        for row in rows:
            raw2real, time2freq, mix1, conv, mix2, freq2time, real2raw, total = (float(x) for x in row[1:9])
            assert raw2real == 0 and mix2 == 0 and real2raw == 0, row
            assert 0 < time2freq < 5 and 0 < conv < 5 and 0 < freq2time < 5 and total > 0, row
        # the two-input filter's ring fill (mixscale1) runs in the process that owns the last output only
        assert len({row[0] for row in rows if float(row[3]) > 0}) == 1


def _channel_case(seed):
    """test_gpu_fuzz's random N:1 cases (maps, integer delays in every regime of delay.c, run-time delay
    changes and mutes), on integer outputs with dither on some physical outputs for every third seed and
    sub-sample delays on some channels for every fourth"""
    c = fuzz._vchan_case(seed)
    rng = np.random.default_rng(seed + 77)
    n_phys_out = max(c["maps"][1]) + 1
    c["outfmt"] = ["S16_LE", "S24_4LE", cases_float(c["rs"])][seed % 3]
    c["dither"] = [int(rng.random() < 0.6) if c["outfmt"] == "S16_LE" else 0 for _ in range(n_phys_out)]
    c["sdf_length"], c["subdelay"] = -1, [[UNDEF_SUBDELAY] * c["nv"][io] for io in range(2)]
    if seed % 4 == 3:
        c["sdf_length"] = int(rng.choice([3, 7, 15]))
        if 2 * c["sdf_length"] + 1 > c["L"]:
            c["sdf_length"] = 3
        for io in range(2):
            for v in range(c["nv"][io]):
                if rng.random() < 0.5:
                    c["subdelay"][io][v] = int(rng.integers(-99, 100))
        # A channel that shares a physical one and has no filter of its own is delayed by sdf_length more,
        # and the reference adds sdf_length to its maxdelay too (bfrun.c:1152-1162, 1185-1197) -- also to
        # the default -1, which makes it a limit BELOW the delay; delay.c:357-374 then allocates for the
        # limit and fills for the delay.  The reference overruns its heap there (this test found it:
        # "corrupted double-linked list" in the unpatched host); such channels get a real maxdelay here,
        # and test_gpu_features holds the engine to its own defined answer for them.
        for io in range(2):
            if not any(sd != UNDEF_SUBDELAY for sd in c["subdelay"][io]):
                continue
            for v in range(c["nv"][io]):
                shared = c["maps"][io].count(c["maps"][io][v]) > 1
                if shared and c["subdelay"][io][v] == UNDEF_SUBDELAY and c["maxd"][io][v] < 0:
                    c["maxd"][io][v] = 1200
        # run-time changes of the sub-sample value on channels that have a filter (what bfaccess->set_subdelay
        # leaves in icomm->subdelay, bfrun.c:520-541; +-100 and beyond = "no filter this period", delay.c:438-442)
        for _ in range(int(rng.integers(0, 5))):
            io = int(rng.integers(0, 2))
            with_filter = [v for v in range(c["nv"][io]) if c["subdelay"][io][v] != UNDEF_SUBDELAY]
            if with_filter:
                b = int(rng.integers(1, c["n_blocks"]))
                c["events"].setdefault(b, []).append(("subdelay", io, int(rng.choice(with_filter)), int(rng.integers(-99, 100))))
    # channels muted from the start (`mute: true` in the device section -> bfconf->mute -> icomm->ismuted, bfrun.c:2204-2208)
    c["mute"] = [[int(rng.random() < 0.12) for _ in range(c["nv"][io])] for io in range(2)]
    return c


def cases_float(rs):
    return "FLOAT_LE" if rs == 4 else "FLOAT64_LE"


def _channel_engine(cls, c):
    e = cls(c["L"], c["N"], c["rs"], c["nv"][0], c["nv"][1])
    e.map_channels(0, c["maps"][0])
    e.map_channels(1, c["maps"][1])
    e.set_interleaved_phys(0, c["infmt"], max(c["maps"][0]) + 1)
    e.set_interleaved_phys(1, c["outfmt"], max(c["maps"][1]) + 1)
    if c["sdf_length"] > 0:
        e.enable_subdelay(c["sdf_length"], 9.0)
    if any(c["dither"]):
        e.enable_dither([p for p, d in enumerate(c["dither"]) if d], 44100)
    for h in c["coeffs"]:
        e.add_coeff(h)
    for io in range(2):
        for v in range(c["nv"][io]):
            e.set_delay(io, v, c["delay"][io][v])
            e.set_maxdelay(io, v, c["maxd"][io][v])
            if c["mute"][io][v]:
                e.set_mute(io, v, 1)
            if c["sdf_length"] > 0 and c["subdelay"][io][v] != UNDEF_SUBDELAY:
                e.set_subdelay(io, v, c["subdelay"][io][v])
    for f in c["filters"]:
        e.add_filter(**f)
    if hasattr(e, "finalize"):
        e.finalize()
    return e


@pytest.mark.parametrize("seed", range(int(os.environ.get("BFHIP_REFLOOP_SEED0", "0")),
                                        int(os.environ.get("BFHIP_REFLOOP_SEED0", "0")) + int(os.environ.get("BFHIP_REFLOOP_SEEDS", "24"))))
def test_reference_filter_process_with_shared_channels_delays_mutes_dither_subdelay(hip, tmp_path, seed):
    """bfrun.c:1128-1198, 1505-1531, 1938-2003 -- virtual channels that share a physical one are delayed
    (the reference's delay.c), muted and mixed inside filter_process(); the HP-TPDF dither of the
    reference's dither.c and its sub-sample delay filters (delay.c:416-505, over the product's
    convolver_td_* ops) ride along.  The reference's loop over the product's per-block ops, the PATCHED
    loop (fused engine built by bfhip_setup() from the same bfconf), the engine driven directly and the
    oracle, on the same random cases as test_gpu_fuzz's channel test."""
    for exe in (EXE, EXE_PATCHED):
        if not os.path.exists(exe):
            pytest.fail("%s is missing (built from the reference's bfrun.c in the build container)" % exe)
    c = _channel_case(seed)
    n_phys_in, n_phys_out = max(c["maps"][0]) + 1, max(c["maps"][1]) + 1
    spec = dict(L=c["L"], N=c["N"], rs=c["rs"], n_in=c["nv"][0], n_out=c["nv"][1], infmt=c["infmt"], outfmt=c["outfmt"],
                coeffs=[(h, 1.0, 0) for h in c["coeffs"]], filters=[_f(**f) for f in c["filters"]],
                channels=dict(maps=c["maps"], delay=c["delay"], maxdelay=c["maxd"], mute=c["mute"],
                              subdelay=c["subdelay"], dither=c["dither"], sdf_length=c["sdf_length"]))
    blocks = cases.raw_blocks(seed, c["n_blocks"], c["L"], n_phys_in, c["infmt"], amplitude=0.2)
    plain, plain_of, _ = _run_host(EXE, tmp_path, "plain", spec, blocks, c["events"])
    fused, fused_of, _ = _run_host(EXE_PATCHED, tmp_path, "fused", spec, blocks, c["events"])
    odt = {"FLOAT_LE": np.float32, "FLOAT64_LE": np.float64, "S24_4LE": np.int32, "S16_LE": np.int16}[c["outfmt"]]
    ref = np.frombuffer(plain, odt).reshape(c["n_blocks"], c["L"], n_phys_out).astype(np.float64)
    pat = np.frombuffer(fused, odt).reshape(c["n_blocks"], c["L"], n_phys_out).astype(np.float64)
    ge, oe = _channel_engine(hip.Engine, c), _channel_engine(bo.Engine, c)
    undefined = _undefined_dither_samples(c["L"], c["rs"], sum(c["dither"]), c["n_blocks"]) if any(c["dither"]) else None
    tol = 3e-5 if c["rs"] == 4 else 1e-11
    scale = float(np.abs(ref).max())
    for b, blk in enumerate(blocks):
        for eng in (ge, oe):
            for kind, io, v, val in c["events"].get(b, []):
                {"delay": eng.set_delay, "mute": eng.set_mute, "subdelay": eng.set_subdelay}[kind](io, v, val)
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0, (seed, b)
        got = {"patched host": pat[b],
               "fused engine": np.frombuffer(g.tobytes(), odt).reshape(c["L"], n_phys_out).astype(np.float64),
               "oracle": np.frombuffer(o.tobytes(), odt).reshape(c["L"], n_phys_out).astype(np.float64)}
        for who, arr in got.items():
            if c["outfmt"] in ("S16_LE", "S24_4LE"):
                # one count: a sample on a rounding boundary.  A dithered channel may differ by two: the
                # HP-TPDF quantiser feeds its error back with coefficients {1, -1} (dither_funs.h:20-27), so
                # one rounding decision that falls the other way moves the next two samples by a count
                # each way on top of their own rounding
                # (... plus, on 24-bit outputs of a float32 engine, the float tolerance in counts: a sub-sample FIR,
                # a mix and the transforms' round trips at a level of a million counts -- seed 3187: 5 counts)
                full = 32768.0 if c["outfmt"] == "S16_LE" else 8388608.0
                dith = np.asarray(c["dither"], bool)
                dev = np.abs(arr - ref[b])
                for k, pch in enumerate(np.nonzero(dith)[0]):         # (where the reference reads past its dither table)
                    if undefined[k][b]:
                        dev[sorted(undefined[k][b]), pch] = 0.0
                d = dev.max(axis=0)
                for pch in range(n_phys_out):
                    lvl = max(float(np.sqrt((ref[b][:, pch] ** 2).mean())), 1e-3 * scale, fuzz.FLOOR * full)
                    lim = (2.0 if dith[pch] else 1.0) + 16 * tol * lvl
                    assert d[pch] <= lim, (who, seed, b, pch, float(d[pch]), lim)
                continue
            else:
                for ch in range(n_phys_out):
                    lvl = max(float(np.sqrt((ref[b][:, ch] ** 2).mean())), 1e-3 * scale, fuzz.FLOOR)
                    err = float(np.sqrt(((arr[:, ch] - ref[b][:, ch]) ** 2).mean()))
                    assert err <= tol * lvl, (who, seed, b, ch, err, lvl)
    assert len(plain_of) == len(fused_of) == c["nv"][1]
    # ... and dealt out over two and three filter processes (the members of a shared physical output in
    # one process, bfconf.c:2893-2931): the same bytes as one process, patched and unpatched
    from test_gpu_shards import _assign
    by_phys = dict(spec, filters=[dict(f, out_ch=[c["maps"][1][o] for o in f["out_ch"]]) for f in spec["filters"]])
    rng = np.random.default_rng(seed)
    for n_proc in (2, 3):
        f_owner, _ = _assign(by_phys, n_proc, rng)
        if len(set(f_owner)) != n_proc:
            continue
        many, many_of, _ = _run_host(EXE_PATCHED, tmp_path, "p%d" % n_proc, spec, blocks, c["events"], f_owner)
        assert many == fused and many_of == fused_of, (seed, n_proc, f_owner)
        ref_many, ref_many_of, _ = _run_host(EXE, tmp_path, "u%d" % n_proc, spec, blocks, c["events"], f_owner)
        assert ref_many == plain and ref_many_of == plain_of, (seed, n_proc, f_owner)


@pytest.mark.parametrize("rs", [4, 8])
@pytest.mark.parametrize("mode", ["exact", "floor"])
def test_reference_filter_process_with_powersave(hip, tmp_path, rs, mode):
    """`powersave:` through the reference's loop (test_silent, bfrun.c:721-771; silent windows get a zero
    spectrum, :1541-1553; filters whose inputs have been silent for a whole filter length are skipped and
    their output buffers zeroed, :1694-1770): inputs go silent, come back, hover around the floor, one
    never carries anything.  Unpatched host, patched host, the engine and the oracle."""
    L, N, I, O = 128, 4, 3, 2
    rng = np.random.default_rng(300)
    coeffs = [(cases.make_ir(rng, L * N, I), 1.0, 0) for _ in range(I * O)]
    filters = [_f(in_ch=[i], out_ch=[o], coeff=o * I + i, delayblocks=i % 2) for o in range(O) for i in range(I)]
    filters.append(_f(in_ch=[0, 1], in_scale=[0.5, 0.5], out_ch=[0], coeff=0))
    ofmt = "FLOAT_LE" if rs == 4 else "FLOAT64_LE"
    spec = dict(L=L, N=N, rs=rs, n_in=I, n_out=O, infmt="S16_LE", outfmt=ofmt, coeffs=coeffs, filters=filters)
    thr = 1.0 if mode == "exact" else 10 ** (-50 / 20)
    amps = [3000, 3000, 0, 0, 0, 0, 0, 0, 40, 40, 200, 0, 3000, 0, 0, 0, 0, 0, 0, 3000]
    blocks = []
    for a in amps:
        x = np.zeros((L, I), np.int16)
        x[:, 0] = (rng.standard_normal(L) * a).astype(np.int16)
        x[:, 1] = (rng.standard_normal(L) * (a // 2)).astype(np.int16)
        blocks.append(x)
    env = {"BFREF_POWERSAVE": repr(thr)}
    plain, _, _ = _run_host(EXE, tmp_path, "plain", spec, blocks, {}, env=env)
    fused, _, _ = _run_host(EXE_PATCHED, tmp_path, "fused", spec, blocks, {}, env=env)
    off, _, _ = _run_host(EXE, tmp_path, "off", spec, blocks, {})
    if mode == "exact":
        assert plain == off                      # the reference itself: `powersave: true` changes no sample
    odt = np.float32 if rs == 4 else np.float64
    ref = np.frombuffer(plain, odt).reshape(len(amps), L, O).astype(np.float64)
    pat = np.frombuffer(fused, odt).reshape(len(amps), L, O).astype(np.float64)

    def mk(cls):
        e = cls(L, N, rs, I, O)
        e.set_interleaved(0, "S16_LE")
        e.set_interleaved(1, ofmt)
        e.set_powersave(thr)
        for t, s_, nb in coeffs:
            e.add_coeff(t, s_, nb)
        for f in filters:
            e.add_filter(**f)
        if hasattr(e, "finalize"):
            e.finalize()
        return e
    ge, oe = mk(hip.Engine), mk(bo.Engine)
    scale = float(np.abs(ref).max())
    tol = (3e-5 if rs == 4 else 1e-11) * scale
    for b, blk in enumerate(blocks):
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        for who, arr in (("patched host", pat[b]), ("fused engine", np.frombuffer(g.tobytes(), odt).reshape(L, O)),
                         ("oracle", np.frombuffer(o.tobytes(), odt).reshape(L, O))):
            assert np.abs(arr.astype(np.float64) - ref[b]).max() <= tol, (who, b)


@pytest.mark.parametrize("rs", [4, 8])
def test_coefficient_partitions_rewritten_in_shared_memory_while_the_host_runs(hip, tmp_path, rs):
    """`shared_mem: true;` coefficient sets rewritten in place by another process while the filter
    process runs -- what bflogic_eq does after rendering a curve (convolver_runtime_coeffs2cbuf through
    bfaccess, bfrun.c:2305).  The unpatched host reads the partition from shared memory in the next
    block; the patched host registered the set BFHIP_COEFF_WATCH and has to notice by itself, in the
    same block.  Against the engine driven directly (bfhip_engine_update_coeff_block)."""
    L, N, I, O = 256, 4, 2, 2
    rng = np.random.default_rng(17)
    dt = np.float32 if rs == 4 else np.float64
    taps = [cases.make_ir(rng, L * N, I).astype(dt) for _ in range(3)]
    scales = [1.0, -0.5, 2.0]
    coeffs = [(t, sc, 0) for t, sc in zip(taps, scales)]
    filters = [_f(in_ch=[0], out_ch=[0], coeff=0), _f(in_ch=[1], out_ch=[1], coeff=1), _f(in_ch=[0, 1], in_scale=[0.5, -0.5], out_ch=[1], coeff=2),
               _f(in_ch=[1], out_ch=[0], coeff=0, delayblocks=1)]
    ofmt = "FLOAT_LE" if rs == 4 else "FLOAT64_LE"
    spec = dict(L=L, N=N, rs=rs, n_in=I, n_out=O, infmt="S24_4LE", outfmt=ofmt, coeffs=coeffs, filters=filters)
    events = {3: [("rewrite", 0, 1, 0.25)], 5: [("rewrite", 2, 0, -1.5), ("rewrite", 2, 3, 0.0)], 6: [("rewrite", 0, 1, 1.0)],
              9: [("rewrite", 1, 2, 3.0), ("coeff", 1, 2)]}
    blocks = cases.raw_blocks(8, 14, L, I, "S24_4LE", amplitude=0.2)
    env = {"BFREF_SHARED_COEFFS": "1"}
    plain, _, _ = _run_host(EXE, tmp_path, "plain", spec, blocks, events, env=env)
    fused, _, _ = _run_host(EXE_PATCHED, tmp_path, "fused", spec, blocks, events, env=env)
    still, _, _ = _run_host(EXE, tmp_path, "still", spec, blocks, {9: [("coeff", 1, 2)]}, env=env)
    assert plain != still                        # the rewrites are audible
    # two filter processes, every set registered lazily and watched in both: the same bytes
    two, _, _ = _run_host(EXE_PATCHED, tmp_path, "fused2", spec, blocks, events, [0, 1, 1, 0], env=env)
    assert two == fused
    ref = np.frombuffer(plain, dt).reshape(len(blocks), L, O).astype(np.float64)
    pat = np.frombuffer(fused, dt).reshape(len(blocks), L, O).astype(np.float64)
    ge = cases.build(hip.Engine, spec)
    tol = (3e-5 if rs == 4 else 1e-11) * float(np.abs(ref).max())
    for b, blk in enumerate(blocks):
        for ev in events.get(b, []):
            if ev[0] == "rewrite":
                _, cs, part, gain = ev
                ge.update_coeff_block(cs, part, (taps[cs][part * L:(part + 1) * L].astype(np.float64) * scales[cs] * gain).astype(dt))
            else:
                ge.set_coeff(ev[1], ev[2])
        st, g = ge.block(blk)
        assert st == 0
        eng = np.frombuffer(g.tobytes(), dt).reshape(L, O).astype(np.float64)
        assert np.abs(pat[b] - ref[b]).max() <= tol, ("patched host", b, float(np.abs(pat[b] - ref[b]).max()))
        assert np.abs(eng - ref[b]).max() <= tol, ("engine", b)


def test_filter_processes_keep_step_on_their_shared_wake_pipe(hip, tmp_path):
    """All filter processes are woken through ONE pipe, n_processes tokens per period
    (bfrun.c:2480-2616); in the reference a synch_filter_processes() barrier keeps a fast process from
    taking a second token of the same period.  The fused path needs no data from the other processes,
    but it needs that barrier: with process 0 held back while the tokens go out (BFREF_STALL_PROC0), a
    host without it lets another process run process 0's period on a buffer nobody has filled yet.
    (Found by a soak: one three-process run in some thousand wrote a wrong block.)"""
    L, N, I, O = 256, 3, 3, 3
    rng = np.random.default_rng(3)
    spec = dict(L=L, N=N, rs=4, n_in=I, n_out=O, infmt="S24_4LE", outfmt="S24_4LE",
                coeffs=[(cases.make_ir(rng, L * N, I), 1.0, 0) for _ in range(3)],
                filters=[_f(in_ch=[i], out_ch=[o], coeff=(i + o) % 3) for o in range(O) for i in range(I)])
    blocks = cases.raw_blocks(1, 32, L, I, "S24_4LE", amplitude=0.2)     # (the first periods are slow ones: set-up, graph capture)
    f_owner = [f["out_ch"][0] for f in spec["filters"]]          # process k mixes output k
    one, one_of, _ = _run_host(EXE_PATCHED, tmp_path, "one", spec, blocks, {})
    for exe, tag in ((EXE_PATCHED, "patched"), (EXE, "plain")):
        want = one if exe == EXE_PATCHED else _run_host(EXE, tmp_path, "plain1", spec, blocks, {})[0]
        got, _, _ = _run_host(exe, tmp_path, tag + "3", spec, blocks, {}, f_owner, env={"BFREF_STALL_PROC0": "40000"})     # 40 ms a period: many periods of the others, also on a shared GPU
        assert got == want, tag


def _combined_case(seed):
    """a random filter NETWORK (cascades, cross-fades, delayblocks, run-time control: test_gpu_fuzz._network)
    on top of random channel maps (N:1 both sides, integer delays, mutes, and for some seeds dither or
    sub-sample delays): the two generators the other tests use apart, together"""
    # (every seventh case at the partition lengths of the wave-level transforms, fft_wave.h: 4096 by default)
    spec, n_blocks, events = fuzz._network(seed + 20000, Ls=(4096, 8192), max_n=3) if seed % 7 == 6 else fuzz._network(seed + 20000)
    rng = np.random.default_rng(seed + 555)
    nv = [spec["n_in"], spec["n_out"]]
    maps = []
    for io in range(2):
        n_phys = int(rng.integers(1, nv[io] + 1))
        v2p = list(range(n_phys)) + [int(rng.integers(0, n_phys)) for _ in range(nv[io] - n_phys)]     # every physical channel used
        maps.append([int(x) for x in rng.permutation(v2p)])
    maxd = [[int(rng.choice([0, 40, 300, 900, -1])) for _ in range(nv[io])] for io in range(2)]
    lim = lambda md: 1200 if md < 0 else md        # noqa: E731
    delay = [[int(rng.integers(0, lim(maxd[io][v]) + 1)) for v in range(nv[io])] for io in range(2)]
    outfmt = ["S16_LE", "S24_4LE", cases_float(spec["rs"])][seed % 3]
    n_phys_out = max(maps[1]) + 1
    dither = [int(rng.random() < 0.6) if outfmt == "S16_LE" else 0 for _ in range(n_phys_out)]
    sdf_length, subdelay = -1, [[UNDEF_SUBDELAY] * nv[io] for io in range(2)]
    if seed % 4 == 3:
        sdf_length = int(rng.choice([3, 7, 15]))
        for io in range(2):
            for v in range(nv[io]):
                if rng.random() < 0.4:
                    subdelay[io][v] = int(rng.integers(-99, 100))
        for io in range(2):                          # (see _channel_case: where the reference overruns its own buffer)
            if any(sd != UNDEF_SUBDELAY for sd in subdelay[io]):
                for v in range(nv[io]):
                    if maps[io].count(maps[io][v]) > 1 and subdelay[io][v] == UNDEF_SUBDELAY and maxd[io][v] < 0:
                        maxd[io][v] = 1200
    ch_events = {}
    for _ in range(int(rng.integers(0, 8))):
        b = int(rng.integers(1, n_blocks))
        io = int(rng.integers(0, 2))
        v = int(rng.integers(0, nv[io]))
        if rng.random() < 0.6:
            ch_events.setdefault(b, []).append(("delay", io, v, int(rng.integers(0, lim(maxd[io][v]) + 1))))
        else:
            ch_events.setdefault(b, []).append(("mute", io, v, int(rng.integers(0, 2))))
    mute = [[int(rng.random() < 0.1) for _ in range(nv[io])] for io in range(2)]
    spec = dict(spec, outfmt=outfmt)
    # `powersave: true;` (exact zeros) for every fifth case, with stretches of silence in the input.  It is
    # meant to skip work, not to change samples (brutefir.html: "pause filtering when input is zero"), and in
    # this engine it does not; the reference's zero-flag bookkeeping does in three corners (a
    # run-time `delayblocks` change -- a burst of 0.5 on a 0.03 signal in seed 14; a cross-fading switch while
    # some partitions are flagged silent -- crossfadebuf mixed in unwritten, "NaN or Inf" aborts; a silent
    # member of a shared physical output behind a delayed one -- ocbuf[0] is delayed in place and still
    # flagged zero, seed 109).  So these cases are held to the reference's run WITHOUT powersave.
    powersave = 1.0 if seed % 5 == 4 else 0.0
    return dict(spec=spec, n_blocks=n_blocks, events=events, ch_events=ch_events, maps=maps, nv=nv, maxd=maxd, delay=delay,
                dither=dither, sdf_length=sdf_length, subdelay=subdelay, mute=mute, powersave=powersave)


def _undefined_dither_samples(L, rs, n_dithered, n_blocks):
    """The HP-TPDF quantiser looks its dither up at randmap[r[n] - r[n-1]], r int8: the index runs to +255,
    the reference's table to +254 (dither.c:115-130) -- what it adds for 127 - (-128) is whatever lies
    behind its allocation (this engine continues the table's formula).  One sample in 65536:
    with partitions of thousands of samples some block of a run meets it.  The walk through the random
    table is reproduced here (the oracle's table is bit-identical to dither.c's, tests/test_oracle_golden.py)
    and those samples are named, per dithered channel (in the order of the dither list) and block -- with
    the 31 behind each: the error feedback carries a stray value two samples on, and through as many clipped
    samples as it takes to come back into range when the stray value was huge (seed 153: 7 samples)."""
    ctx = bo.Ctx(L, rs)
    if not ctx.dither_init(n_dithered, 44100):
        return None
    tab = ctx.dither_table().astype(np.int64)
    ptr = [int(bo.lib().bfo_dither_randtab_ptr(ctx.h, k)) for k in range(n_dithered)]
    bad = [[set() for _ in range(n_blocks)] for _ in range(n_dithered)]
    for b in range(n_blocks):
        for k in range(n_dithered):
            p = ptr[k]
            if p + L >= len(tab):                       # dither_preloop_real2int_hp_tpdf, dither.h:28-38
                tab[0] = tab[p - 1]
                p = 1
            seg = tab[p - 1:p + L]
            for i in np.nonzero(seg[1:] - seg[:-1] == 255)[0]:
                bad[k][b].update(range(int(i), min(int(i) + 32, L)))
            ptr[k] = p + L
    return bad


def _combined_engine(cls, c):
    spec = c["spec"]
    e = cls(spec["L"], spec["N"], spec["rs"], spec["n_in"], spec["n_out"])
    e.map_channels(0, c["maps"][0])
    e.map_channels(1, c["maps"][1])
    e.set_interleaved_phys(0, spec["infmt"], max(c["maps"][0]) + 1)
    e.set_interleaved_phys(1, spec["outfmt"], max(c["maps"][1]) + 1)
    if c["powersave"]:
        e.set_powersave(c["powersave"])
    if c["sdf_length"] > 0:
        e.enable_subdelay(c["sdf_length"], 9.0)
    if any(c["dither"]):
        e.enable_dither([p for p, d in enumerate(c["dither"]) if d], 44100)
    for taps, scale, nb in spec["coeffs"]:
        e.add_coeff(taps, scale, nb)
    for io in range(2):
        for v in range(c["nv"][io]):
            e.set_delay(io, v, c["delay"][io][v])
            e.set_maxdelay(io, v, c["maxd"][io][v])
            if c["mute"][io][v]:
                e.set_mute(io, v, 1)
            if c["sdf_length"] > 0 and c["subdelay"][io][v] != UNDEF_SUBDELAY:
                e.set_subdelay(io, v, c["subdelay"][io][v])
    for f in spec["filters"]:
        e.add_filter(**f)
    if hasattr(e, "finalize"):
        e.finalize()
    return e


@pytest.mark.parametrize("seed", range(int(os.environ.get("BFHIP_REFLOOP_SEED0", "0")),
                                        int(os.environ.get("BFHIP_REFLOOP_SEED0", "0")) + int(os.environ.get("BFHIP_REFLOOP_SEEDS", "24"))))
def test_reference_filter_process_on_random_networks_over_random_channel_maps(hip, tmp_path, seed):
    """everything at once: filter networks with run-time control (cascades through convolve_eval,
    cross-fades, coeff -1, delayblocks) over N:1 channel maps with delays, mutes, dither and sub-sample
    delays.  The reference's loop (unpatched), the patched loop -- which takes the fused path or, where
    bfhip_wanted() says so, the unfused one -- the engine driven directly and the oracle."""
    for exe in (EXE, EXE_PATCHED):
        if not os.path.exists(exe):
            pytest.fail("%s is missing (built from the reference's bfrun.c in the build container)" % exe)
    c = _combined_case(seed)
    spec = c["spec"]
    n_blocks, L = c["n_blocks"], spec["L"]
    n_phys_in, n_phys_out = max(c["maps"][0]) + 1, max(c["maps"][1]) + 1
    hspec = dict(spec, channels=dict(maps=c["maps"], delay=c["delay"], maxdelay=c["maxd"], mute=c["mute"],
                                     subdelay=c["subdelay"], dither=c["dither"], sdf_length=c["sdf_length"]))
    events = {b: list(c["events"].get(b, [])) + list(c["ch_events"].get(b, [])) for b in set(c["events"]) | set(c["ch_events"])}
    blocks = cases.raw_blocks(seed, n_blocks, L, n_phys_in, spec["infmt"], amplitude=0.2)
    env = {}
    if c["powersave"]:
        env["BFREF_POWERSAVE"] = repr(c["powersave"])
        prng = np.random.default_rng(seed + 99)
        for blk in blocks:                               # stretches of silence and of near silence
            for ch in range(n_phys_in):
                u = prng.random()
                if u < 0.3:
                    blk[:, ch] = 0
                elif u < 0.4 and not spec["infmt"].startswith("FLOAT"):
                    blk[:, ch] = (blk[:, ch].astype(np.int64) >> 12).astype(blk.dtype)
    plain, _, _ = _run_host(EXE, tmp_path, "plain", hspec, blocks, events)               # (without powersave: see _combined_case)
    fused, _, _ = _run_host(EXE_PATCHED, tmp_path, "fused", hspec, blocks, events, env=env)
    odt = {"FLOAT_LE": np.float32, "FLOAT64_LE": np.float64, "S24_4LE": np.int32, "S16_LE": np.int16}[spec["outfmt"]]
    ref = np.frombuffer(plain, odt).reshape(n_blocks, L, n_phys_out).astype(np.float64)
    pat = np.frombuffer(fused, odt).reshape(n_blocks, L, n_phys_out).astype(np.float64)
    ge, oe = _combined_engine(hip.Engine, c), _combined_engine(bo.Engine, c)
    dith_rank = {p: k for k, p in enumerate(p for p, d in enumerate(c["dither"]) if d)}
    undefined = _undefined_dither_samples(L, spec["rs"], len(dith_rank), n_blocks) if dith_rank else None
    tol = 3e-5 if spec["rs"] == 4 else 1e-11
    full = {"S16_LE": 32768.0, "S24_4LE": 8388608.0}.get(spec["outfmt"], 1.0)
    scale = float(np.abs(ref).max())
    for b, blk in enumerate(blocks):
        for eng in (ge, oe):
            fuzz._apply(eng, c["events"].get(b, []))
            for kind, io, v, val in c["ch_events"].get(b, []):
                {"delay": eng.set_delay, "mute": eng.set_mute}[kind](io, v, val)
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0, (seed, b)
        got = {"patched host": pat[b],
               "fused engine": np.frombuffer(g.tobytes(), odt).reshape(L, n_phys_out).astype(np.float64),
               "oracle": np.frombuffer(o.tobytes(), odt).reshape(L, n_phys_out).astype(np.float64)}
        for who, arr in got.items():
            for ch in range(n_phys_out):
                want = ref[b][:, ch]
                if full > 1.0:
                    # integer outputs: the float tolerance in counts (a cross-fade block is three more float32
                    # round trips), plus the rounding itself -- two counts on a dithered channel (error feedback)
                    lvl = max(float(np.sqrt((want ** 2).mean())), 1e-3 * scale, fuzz.FLOOR * full)
                    lim = (2.0 if c["dither"][ch] else 1.0) + 4 * tol * lvl * 4
                    dev = np.abs(arr[:, ch] - want)
                    if c["dither"][ch] and undefined[dith_rank[ch]][b]:
                        dev[sorted(undefined[dith_rank[ch]][b])] = 0.0          # the reference read past its table there
                    assert dev.max() <= lim, (who, seed, b, ch, float(dev.max()), lim)
                else:
                    lvl = max(float(np.sqrt((want ** 2).mean())), 1e-3 * scale, fuzz.FLOOR)
                    err = float(np.sqrt(((arr[:, ch] - want) ** 2).mean()))
                    assert err <= tol * lvl, (who, seed, b, ch, err, lvl)
    # ... and the patched host with the filters dealt out over two and three filter processes (connected
    # filters and the members of a shared physical output together): the bytes of its one-process run
    from test_gpu_shards import _assign
    by_phys = dict(spec, filters=[dict(f, out_ch=[c["maps"][1][o] for o in f["out_ch"]]) for f in spec["filters"]])
    rng = np.random.default_rng(seed)
    for n_proc in (2, 3):
        f_owner, _ = _assign(by_phys, n_proc, rng)
        if len(set(f_owner)) != n_proc:
            continue
        many, _, _ = _run_host(EXE_PATCHED, tmp_path, "p%d" % n_proc, hspec, blocks, events, f_owner, env=env)
        assert many == fused, (seed, n_proc, f_owner)


def test_two_periods_in_flight_are_opt_in_and_for_blocking_io_only(hip, tmp_path):
    """The patched host keeps the reference's I/O delay (brutefir.html:839) by default; BFHIP_TWO_PERIODS=1
    trades one more period of delay for the overlap of copies and kernels: the same samples, one block
    later.  With callback I/O (JACK) the switch is ignored: the callback must get its period back."""
    for exe in (EXE, EXE_PATCHED):
        if not os.path.exists(exe):
            pytest.fail("%s is missing (built from the reference's bfrun.c in the build container)" % exe)
    spec, n_blocks, events = fuzz._network(9003)
    spec = dict(spec, outfmt="S24_4LE")
    blocks = cases.raw_blocks(3, n_blocks, spec["L"], spec["n_in"], spec["infmt"], amplitude=0.2)
    blk_bytes = spec["L"] * spec["n_out"] * 4
    normal, _, _ = _run_host(EXE_PATCHED, tmp_path, "normal", spec, blocks, events)
    late, _, _ = _run_host(EXE_PATCHED, tmp_path, "late", spec, blocks, events, env={"BFHIP_TWO_PERIODS": "1"})
    assert late[blk_bytes:] == normal[:-blk_bytes] and late != normal
    cb, _, _ = _run_host(EXE_PATCHED, tmp_path, "cb", spec, blocks, events, env={"BFREF_CALLBACK_IO": "1", "BFHIP_TWO_PERIODS": "1"})
    assert cb == normal
    plain_cb, _, _ = _run_host(EXE, tmp_path, "plain_cb", spec, blocks, events, env={"BFREF_CALLBACK_IO": "1"})
    plain, _, _ = _run_host(EXE, tmp_path, "plain", spec, blocks, events)
    assert plain_cb == plain                     # the reference's own loop: the same bytes on either kind of pipe
