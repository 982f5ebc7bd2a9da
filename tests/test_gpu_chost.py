"""GPU: the C ABI driven from plain C (examples/bffilter.c, gcc, no Python in the data path):
file -> file filtering like `brutefir` with bfio_file on both sides; output compared with the
oracle run on the same file."""
import os
import subprocess

import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("L,N", [(512, 4), (65536, 1)])
def test_c_host_file_to_file(hip, tmp_path, L, N):
    exe = os.path.join(ROOT, "examples", "bffilter")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "bffilter.c"), "-o", exe,
                           "-L" + os.path.join(ROOT, "brutefir_amd"), "-lbfhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "brutefir_amd")])
    I, O = 2, 3                                  # (65536, 1): the stock `filter_length: 65536;`
    irs = np.stack([cases.make_ir(np.random.default_rng(100 + k), L * N, I) for k in range(O * I)])
    irs.astype(np.float32).tofile(tmp_path / "coeffs.f32")
    nblk = 9 if L < 4096 else 3
    blocks = cases.raw_blocks(42, nblk, L, I, "S16_LE", amplitude=0.2)
    raw = np.concatenate(blocks)[:nblk * L - 100]                   # a ragged last block
    raw.tofile(tmp_path / "in.raw")
    r = subprocess.run([exe, str(L), str(N), str(I), str(O), "S16_LE", "S24_4LE",
                        str(tmp_path / "coeffs.f32"), str(tmp_path / "in.raw"),
                        str(tmp_path / "out.raw")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "%d blocks" % nblk in r.stderr
    if L < 4096:
        # the same run with the reference's `benchmark: true` stage table (bfrun.c:2035-2078) filled
        # from bfhip_engine_stage_times(): dithered 16-bit output -> the real2raw column has a time
        rb = subprocess.run([exe, str(L), str(N), str(I), str(O), "S16_LE", "S16_LE",
                             str(tmp_path / "coeffs.f32"), str(tmp_path / "in.raw"),
                             str(tmp_path / "out_bm.raw"), "44100", "benchmark"], capture_output=True, text=True, timeout=120)
        assert rb.returncode == 0, rb.stderr
        # (9 blocks: no tenth period, no row; the header test lives with bfprocs -- here only that timing changes nothing)
        ra = subprocess.run([exe, str(L), str(N), str(I), str(O), "S16_LE", "S16_LE",
                             str(tmp_path / "coeffs.f32"), str(tmp_path / "in.raw"),
                             str(tmp_path / "out_nobm.raw"), "44100"], capture_output=True, text=True, timeout=120)
        assert ra.returncode == 0 and open(tmp_path / "out_bm.raw", "rb").read() == open(tmp_path / "out_nobm.raw", "rb").read()
    got = np.fromfile(tmp_path / "out.raw", np.int32).reshape(-1, O)
    assert got.shape[0] == raw.shape[0]                             # as many frames out as in
    oe = bo.Engine(L, N, 4, I, O)
    oe.set_interleaved(0, "S16_LE")
    oe.set_interleaved(1, "S24_4LE")
    for o in range(O):
        for i in range(I):
            oe.add_filter(in_ch=[i], out_ch=[o], coeff=oe.add_coeff(irs[o * I + i].astype(np.float32)))
    padded = np.concatenate([raw, np.zeros((100, I), np.int16)])
    want = []
    for b in range(nblk):
        _, out = oe.block(padded[b * L:(b + 1) * L])
        want.append(out.view(np.int32).reshape(L, O))
    want = np.concatenate(want)[:raw.shape[0]]
    assert np.abs(got.astype(np.int64) - want).max() <= 1


def test_c_host_realtime_round_trip(hip):
    """examples/bflatency.c: the same periods through bfhip_engine_block and the three real-time
    variants from plain C; outputs must be byte-identical and the replayed path must not be
    slower than the plain one (the latency numbers themselves are in profiles/)."""
    import json
    exe = os.path.join(ROOT, "examples", "bflatency")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "bflatency.c"), "-o", exe,
                           "-L" + os.path.join(ROOT, "brutefir_amd"), "-lbfhip", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "brutefir_amd")])
    r = subprocess.run([exe, "300"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) >= 6
    for ln in lines:
        assert ln["outputs_identical"] is True, ln
        assert ln["rt_spin"][0] <= ln["block"][0] * 1.1, ln


def test_c_host_with_rccl_reduce_scatter(hip):
    """examples/bfmulti.c: the sharded block (inputs -> partial spectra -> ncclReduceScatter ->
    outputs) driven from plain C with RCCL's C API.  On this one-GPU box the communicator has a
    single rank, so the collective is a copy and the result must equal the fused block call
    byte for byte; with more GPUs the same binary shards the inputs over all of them."""
    import json
    exe = os.path.join(ROOT, "examples", "bfmulti")
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "bfmulti.c"),
                           "-o", exe, "-L" + os.path.join(ROOT, "brutefir_amd"), "-lbfhip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "brutefir_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, "1", "12", "8", "8", "1024", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["status_bits"] == 0
    assert line["max_abs_output"] > 1000 and line["max_abs_difference_vs_block_dev"] == 0


def test_c_host_three_processes_hot_swap(hip, tmp_path):
    """examples/bfhotswap.c: BruteFIR's process topology from plain C -- the parent prepares the
    coefficients (host code, never initialises HIP) and forks; the forked FILTER process owns the
    GPU; a forked MODULE process rewrites one partition in shared memory the way bflogic_eq does;
    the filter process picks it up at its next block.  Output against the oracle."""
    exe = os.path.join(ROOT, "examples", "bfhotswap")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "bfhotswap.c"), "-o", exe,
                           "-L" + os.path.join(ROOT, "brutefir_amd"), "-lbfhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "brutefir_amd")])
    L, N, nblk, sw = 1024, 4, 11, 6
    rng = np.random.default_rng(77)
    h0 = cases.make_ir(rng, L * N, 1).astype(np.float32)
    part = cases.make_ir(rng, L, 1).astype(np.float32)
    x = (rng.standard_normal(nblk * L) * 0.1).astype(np.float32)
    h0.tofile(tmp_path / "taps0.f32")
    part.tofile(tmp_path / "part.f32")
    x.tofile(tmp_path / "in.f32")
    r = subprocess.run([exe, str(L), str(N), str(nblk), str(sw), str(tmp_path / "taps0.f32"),
                        str(tmp_path / "part.f32"), str(tmp_path / "in.f32"), str(tmp_path / "out.f32")],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    assert "replaced from block %d" % sw in r.stderr
    got = np.fromfile(tmp_path / "out.f32", np.float32).reshape(nblk, L)
    oe = bo.Engine(L, N, 4, 1, 1)
    oe.set_interleaved(0, "FLOAT_LE")
    oe.set_interleaved(1, "FLOAT_LE")
    h1 = h0.copy()
    h1[L:2 * L] = part
    c0, c1 = oe.add_coeff(h0), oe.add_coeff(h1)
    oe.add_filter(in_ch=[0], out_ch=[0], coeff=c0)
    for b in range(nblk):
        if b == sw:
            oe.set_coeff(0, c1)
        _, o = oe.block(x[b * L:(b + 1) * L].reshape(L, 1))
        assert cases.rel_rms(got[b], np.frombuffer(o.tobytes(), np.float32)) <= 1e-5, b


def _build_bfprocs():
    exe = os.path.join(ROOT, "examples", "bfprocs")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "bfprocs.c"), "-o", exe,
                           "-L" + os.path.join(ROOT, "brutefir_amd"), "-lbfhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "brutefir_amd")])
    return exe


@pytest.mark.parametrize("outfmt", ["S24_4LE", "S16_LE", "FLOAT_LE"])
def test_c_host_forked_filter_processes_write_what_one_process_writes(hip, tmp_path, outfmt):
    """examples/bfprocs.c: the reference's multi-process filter topology (bfrun.c:2312-2328) from
    plain C -- n forked filter processes, one engine each (here all on the one GPU of the box), each
    running the filters of the outputs it owns and writing into the SHARED raw output buffer and
    overflow array (what patches/bfrun-bfhip.diff makes filter_process() do with n_processes > 1).
    The output file and the overflow report of 2 and 3 processes, outputs dealt out interleaved
    (the engine's groups of eight outputs are split between the processes) or in blocks, are
    byte-identical to the one-process run; that run is checked against the oracle."""
    exe = _build_bfprocs()
    L, N, I, O, nblk = 1024, 4, 3, 11, 9
    rng = np.random.default_rng(2026)
    gain = 20.0 if outfmt == "S16_LE" else 1.0          # S16: loud enough to clip now and then
    irs = (np.stack([cases.make_ir(rng, L * N, I) for _ in range(O * I + 1)]) * gain).astype(np.float32)
    irs.tofile(tmp_path / "coeffs.f32")
    amp = 0.9 if outfmt == "S16_LE" else 0.3
    blocks = cases.raw_blocks(7, nblk, L, I, "S24_4LE", amplitude=amp)
    raw = np.concatenate(blocks)[:nblk * L - 300]       # a ragged last block
    raw.tofile(tmp_path / "in.s24")

    def run(n_procs, split):
        out = tmp_path / ("out_%d_%s.raw" % (n_procs, split))
        r = subprocess.run([exe, str(n_procs), split, str(L), str(N), str(I), str(O), outfmt,
                            str(tmp_path / "coeffs.f32"), str(tmp_path / "in.s24"), str(out)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "%d blocks through %d filter process" % (nblk, n_procs) in r.stderr
        return open(out, "rb").read(), r.stdout

    one, report = run(1, "blocked")
    assert len(one) == raw.shape[0] * O * {"S24_4LE": 4, "S16_LE": 2, "FLOAT_LE": 4}[outfmt]      # as many frames out as in
    for n_procs, split in ((2, "interleaved"), (3, "blocked"), (3, "interleaved")):
        got, rep = run(n_procs, split)
        assert got == one, (n_procs, split)
        assert rep == report, (n_procs, split)
    if outfmt == "S16_LE":
        assert any(" 0 overflows" not in ln for ln in report.splitlines()), report     # the counters were exercised
    # ... and the one-process output is right: the oracle on the same file
    oe = bo.Engine(L, N, 4, I, O)
    oe.set_interleaved(0, "S24_4LE")
    oe.set_interleaved(1, outfmt)
    for o in range(O):
        for i in range(I):
            oe.add_filter(in_ch=[i], out_ch=[o], coeff=oe.add_coeff(irs[o * I + i]), delayblocks=(o + i) % 2)
    oe.add_filter(in_ch=[0, I - 1], in_scale=[1.0, -0.5], out_ch=[O - 1], coeff=oe.add_coeff(irs[O * I]),
                  delayblocks=(O - 1 + 1) % 2)
    padded = np.concatenate([raw, np.zeros((300, I), raw.dtype)])
    want = np.concatenate([oe.block(padded[b * L:(b + 1) * L])[1] for b in range(nblk)])
    dt = {"S24_4LE": np.int32, "S16_LE": np.int16, "FLOAT_LE": np.float32}[outfmt]
    w = want.view(dt).reshape(-1, O)[:raw.shape[0]]
    g = np.frombuffer(one, dt).reshape(-1, O)
    if outfmt == "FLOAT_LE":
        assert cases.rel_rms(g, w) <= 1e-5
    else:
        assert np.abs(g.astype(np.int64) - w).max() <= 1


def test_c_host_benchmark_table_has_device_times(hip, tmp_path):
    """`benchmark: true` of the reference prints one line per filter process and ten periods with
    the time of every stage (bfrun.c:2035-2078).  On the GPU path those columns come from
    bfhip_engine_stage_times(): examples/bfprocs.c prints the same table -- time2freq, convolve and
    freq2time are device milliseconds, the stages fused away read 0, total is their sum."""
    exe = _build_bfprocs()
    L, N, I, O, nblk = 4096, 4, 4, 8, 30
    rng = np.random.default_rng(5)
    np.stack([cases.make_ir(rng, L * N, I) for _ in range(O * I + 1)]).astype(np.float32).tofile(tmp_path / "coeffs.f32")
    np.concatenate(cases.raw_blocks(1, nblk, L, I, "S24_4LE")).tofile(tmp_path / "in.s24")
    r = subprocess.run([exe, "2", "blocked", str(L), str(N), str(I), str(O), "S24_4LE", str(tmp_path / "coeffs.f32"),
                        str(tmp_path / "in.s24"), str(tmp_path / "out.raw"), "benchmark"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [[c.strip() for c in ln.split("|")] for ln in r.stderr.splitlines() if ln.count("|") == 9 and "raw2real" not in ln]
    assert len(rows) == 2 * (nblk // 10), r.stderr            # two processes, every ten periods
    assert len({row[0] for row in rows}) == 2                 # two pids
    for row in rows:
        raw2real, time2freq, mix1, conv, mix2, freq2time, real2raw, total = (float(x) for x in row[1:9])
        assert raw2real == 0 and mix2 == 0 and real2raw == 0
        assert time2freq > 0 and conv > 0 and freq2time > 0 and mix1 >= 0
        assert abs(total - (time2freq + mix1 + conv + freq2time)) < 2e-3
        assert total < 5.0                                    # milliseconds, not ticks
    # mixscale1: the ring fill of the two-input filter, in the process that owns the last output only
    assert len({row[0] for row in rows if float(row[3]) > 0}) == 1
