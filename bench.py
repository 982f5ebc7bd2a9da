#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one block of the hot path over one batch of synthetic input: L new samples on
each of the 64 input channels -> forward FFTs -> 64x64 crossbar multiply-accumulate over 32
coefficient partitions per filter (4096 independent 262144-tap IRs, 8 GiB of coefficients in
HBM) -> inverse FFTs -> 64 x L output samples.  Inputs are resident in HBM when the timed region
starts.  N > 1: the crossbar is sharded by input channel (brutefir_amd/sharding.py), one
process per GPU, one RCCL reduce-scatter per block; total work is fixed ("strong" scaling).

`python bench.py --gpus N` needs no outside launcher: the parent starts N fresh rank processes
itself BEFORE it touches the GPU (the way the reference's host forks its own filter processes,
bfrun.c:2312-2328), relays rank 0's line and exits with the ranks' return code.  It also runs
unchanged under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (the
ranks then come from the environment).  The transport of the mix-down is RCCL; anything else
(no RCCL, fewer devices than ranks) is an error, never a silent fallback -- except when
BFHIP_DIST_BACKEND=gloo is set explicitly (rehearsals on one GPU / CPU-side tests).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (I, O, L, N, realsize, raw format)
    "C": (64, 64, 8192, 32, 4, "S24_4LE"),     # BASELINE.json configs[2], the headline
    "B": (8, 8, 8192, 8, 4, "S24_4LE"),        # configs[1]
    # the remaining BASELINE.json configs are parity-test shapes; timing them is informative only
    "D": (256, 256, 8192, 16, 4, "S24_4LE"),   # configs[3] on ONE GPU: 256 one-to-one filters
    "E": (2, 2, 8192, 128, 8, "FLOAT64_LE"),   # configs[4], uniform stand-in, float64
    # what ONE rank of an N-GPU run of config C computes (inputs sharded, all outputs)
    "C2": (32, 64, 8192, 32, 4, "S24_4LE"), "C4": (16, 64, 8192, 32, 4, "S24_4LE"),
    "C8": (8, 64, 8192, 32, 4, "S24_4LE"),
    # the headline crossbar with half / twice the partitions per filter (how much the per-filter
    # start-up of the MAC's load pipeline costs: tools, not bench lines)
    "C16": (64, 64, 8192, 16, 4, "S24_4LE"), "C64": (64, 64, 8192, 64, 4, "S24_4LE"),
    # a float64 crossbar of the headline's byte volume per filter (informative: the f64 MAC path)
    "F": (32, 32, 8192, 32, 8, "FLOAT64_LE"),
}
DIAGONAL = {"D"}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
# verify(): an integer output is the exact value rounded to the nearest count -- half a count -- plus
# what the float32 transforms may add: 1e-6 of the output's peak (a tenth of north_star's 1e-5 RMS
# tolerance; measured ~1.5e-7 of the peak), at least 1e-3 of a count
INTEGER_VERIFY_BAR = (0.5, 1e-3, 1e-6)
# BASELINE.json "metric", verbatim; "value" is its samples/s half, the HBM GB/s half is in "roofline"
BASELINE_METRIC = "filtered samples/sec + achieved HBM GB/s, 64ch\u00d7256k-tap overlap-save"


def synth_ir_dev(torch, seed, taps, n_in, device):
    """white noise * exponential decay, sum|h| = 1/n_in, generated on the GPU"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    h = torch.randn(taps, generator=g, device=device, dtype=torch.float32)
    h *= torch.exp(-torch.arange(taps, device=device, dtype=torch.float32) / (taps / 6.0))
    h /= h.abs().sum() * n_in
    return h


def synth_raw_blocks(torch, n, L, n_ch, device, seed=1234):
    """n different S24_4LE interleaved blocks of seeded noise at -20 dBFS, in HBM"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randn(n, L, n_ch, generator=g, device=device, dtype=torch.float32) * 0.1
    return torch.clamp(torch.round(x * 8388608.0), -8388608, 8388607).to(torch.int32).contiguous()


def _cpu_worker(args):
    """one host core's share of the CPU baseline (runs in a spawned process: no GPU in here)"""
    wl, seed, seconds = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bforacle as bo
    I, O, L, N, rs, fmt = wl
    o_s = min(O, 2)
    e = bo.Engine(L, N, rs, I, o_s)
    for c, f in enumerate(bo.interleaved_formats(fmt, I)):
        e.set_format(0, c, f)
    for c, f in enumerate(bo.interleaved_formats(fmt, o_s)):
        e.set_format(1, c, f)
    e.out_bytes = o_s * L * bo.SAMPLE_FORMATS[fmt][0]
    rng = np.random.default_rng(99 + seed)
    h = rng.standard_normal(L * N).astype(np.float32 if rs == 4 else np.float64)
    h *= np.exp(-np.arange(L * N) / (L * N / 6.0))
    h /= np.abs(h).sum() * I
    t0 = time.time()
    for o in range(o_s):
        for i in range(I):
            c = e.add_coeff(np.roll(h, 17 * (o * I + i)))
            e.add_filter(in_ch=[i], out_ch=[o], coeff=c)
    x = rng.standard_normal((L, I)) * 0.1
    if fmt == "S24_4LE":
        raw = (x * 8388608).astype(np.int32)
    elif fmt == "FLOAT64_LE":
        raw = x.astype(np.float64)
    else:
        raise SystemExit("bench.py: no CPU-baseline input generator for raw format %s" % fmt)
    assert raw.nbytes == L * I * bo.SAMPLE_FORMATS[fmt][0]
    # like the reference (bfrun.c:1745) the oracle only reaches back over blocks that exist:
    # fill the ring first so that every timed block does all N partitions
    for _ in range(N):
        e.block(raw)
    n = 0
    t1 = time.time()
    while True:
        e.block(raw)
        n += 1
        el = time.time() - t1
        if el > seconds or n >= 2000:
            break
    return o_s * L * n / el, n, t1 - t0


def cpu_baseline(wl, seconds=10.0):
    """The oracle (a port of the reference path, oracle/bf_oracle.c) timed on this box's host
    cores on a bounded sample of the same workload: per core a few of the outputs, all inputs,
    all partitions, steady-state blocks after the rings are full.  One process per core, the way
    the reference spreads its filters over `n_processes` (bfconf.c:2227-2318); the single-core
    figure is reported beside the aggregate."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bforacle as bo
    bo.lib()                                          # build once, before the workers race for it
    I, O, L, N, rs, fmt = wl
    # every core this process may really use -- the affinity mask cut down to the cgroup's CPU
    # quota, which is what a container on a shared host actually gets -- and no artificial cap:
    # the reference's default is one filter process per core too (bfconf.c:2227-2318)
    cores, cores_note = usable_cores()
    try:                                              # ... and that the workers' engines fit in memory
        avail = [int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0]
        per_worker = min(O, 2) * I * N * 2 * L * rs * 1.5 + (256 << 20)
        fit = max(1, int(0.5 * avail / per_worker))
        if fit < cores:
            cores_note += ", %d workers fit in half of the available memory" % fit
            cores = fit
    except (OSError, IndexError, ValueError):
        pass
    single = _cpu_worker((wl, 0, seconds / 2))
    agg, per = single[0], [single]
    if cores > 1:
        # plain child processes of this very script (no GPU in them), bounded by a timeout
        name = [k for k, v in WORKLOADS.items() if v == wl][0]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(k),
                                   "--workload", name, "--cpu-seconds", str(seconds)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for k in range(cores)]
        got = []
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=seconds + 120)
                got.append(tuple(json.loads(out.strip().splitlines()[-1])))
            except Exception:                         # noqa: BLE001
                pr.kill()
        if len(got) == cores:
            per, agg = got, sum(g[0] for g in got)
        else:
            cores = 1                                 # report what was measured
    ref_loop = reference_hot_loop(bo, L, rs, I, N, cores=cores)
    return {"value": agg, "unit": "samples/s", "cores": cores, "kind": "port",
            "nproc": os.cpu_count(), "cores_note": cores_note, "cpu_model": cpu_model(),
            "compiler_flags": "gcc -O2 (oracle), gcc -O2 -msse -msse2 (oracle/_ref)",
            "single_core_value": single[0],
            "reference_hot_loop": ref_loop,
            "sample": "per core: %d of %d outputs x %d inputs x %d partitions of %d taps, %d-%d steady-state "
                      "blocks in %.0f s (input FFTs included), gcc -O2, %d worker processes"
                      % (min(O, 2), O, I, N, L, min(p[1] for p in per), max(p[1] for p in per), seconds, cores)}


def _ref_loop_worker(L, rs, seconds, mib):
    """time the reference's own convolve_add (C and SSE builds in oracle/_ref) on this core"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bforacle as bo
    R = bo.ref()
    if R is None:
        return None
    R.ref_set_length(L, 0.0)
    dt = np.float32 if rs == 4 else np.float64
    n_bufs = max(4, (mib << 20) // (2 * L * rs * 2))          # (b, c) pairs, larger than the last-level cache
    rng = np.random.default_rng(1)
    b = rng.standard_normal((n_bufs, 2 * L)).astype(dt)
    c = rng.standard_normal((n_bufs, 2 * L)).astype(dt)
    d = np.zeros(2 * L, dt)
    out = {}
    for name, fn in (("convolve_add_c", R.ref_convolve_add), ("convolve_add_sse", R.ref_convolve_add_simd)):
        calls = 0
        t0 = time.time()
        while time.time() - t0 < seconds / 2:
            for k in range(n_bufs):
                fn(rs, b[k].ctypes.data, c[k].ctypes.data, d.ctypes.data)
            calls += n_bufs
        out[name] = (calls, time.time() - t0)
    return out


def reference_hot_loop(bo, L, rs, I, N, seconds=3.0, cores=1):
    """The reference's OWN inner loop for this path -- convolve_add, the C version its dispatch
    really selects and the SSE version it intends -- compiled from the reference
    sources into oracle/_ref and timed here on cbufs of the workload's size, cycling over a
    working set larger than the last-level cache: on one core, and on all `cores` at once (one
    process per core, the reference's n_processes model).  The full reference binary cannot be
    built (FFTW3 absent); at this workload the loop is > 95 % of the reference's CPU time."""
    import subprocess
    if bo.ref() is None:
        return None
    one = _ref_loop_worker(L, rs, seconds, 512)
    out = {}
    for name, (calls, el) in one.items():
        per_call = el / calls
        # one output sample block (L samples) of one output costs I*N such calls
        out[name] = {"us_per_call": per_call * 1e6, "GB_per_s_3_streams": 3 * 2 * L * rs / per_call / 1e9,
                     "equivalent_samples_per_s": L / (I * N * per_call)}
    if cores > 1:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ref-worker", str(k),
                                   "--ref-shape", "%d,%d" % (L, rs), "--cpu-seconds", str(seconds)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for k in range(cores)]
        got = []
        for pr in procs:
            try:
                o, _ = pr.communicate(timeout=seconds + 120)
                got.append(json.loads(o.strip().splitlines()[-1]))
            except Exception:                         # noqa: BLE001
                pr.kill()
        if len(got) == cores and all(g is not None for g in got):
            for name in one:
                rate = sum(g[name][0] / g[name][1] for g in got)          # calls per second, all cores
                out[name]["all_cores"] = {"cores": cores, "calls_per_s": rate,
                                          "GB_per_s_3_streams": 3 * 2 * L * rs * rate / 1e9,
                                          "equivalent_samples_per_s": L * rate / (I * N)}
    out["note"] = ("gcc -O2 -msse -msse2, %d-byte cbufs; equivalent rate = MAC only, FFTs excluded; "
                   "all_cores = %d processes at once, 128 MiB working set each" % (2 * L * rs, cores))
    return out


def source_hash():
    """sha256 over the device/host sources libbfhip.so is built from: what a committed profile
    has to match to describe the binary that is being timed (there is no .git on the GPU box)"""
    import hashlib
    h = hashlib.sha256()
    files = []
    for d in ("brutefir_amd/csrc", "include"):
        for name in sorted(os.listdir(os.path.join(ROOT, d))):
            if name.endswith((".hip", ".h")):
                files.append(os.path.join(d, name))
    for rel in files:
        h.update(rel.encode())
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def usable_cores():
    """(worker processes to start, how that number came about)"""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:                                              # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        try:                                          # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    n = max(1, min(aff, quota) if quota else aff)
    return n, "affinity mask %d, cgroup cpu quota %s, os.cpu_count %s" % (aff, quota if quota else "none", os.cpu_count())


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _die_with_parent():
    """in the child, before exec: SIGKILL when the launcher goes away (PR_SET_PDEATHSIG), so a rank
    never outlives a launcher that was itself killed at a time limit"""
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, 9)
    except Exception:                                  # noqa: BLE001
        pass


def self_launch(n, argv):
    """`python bench.py --gpus N`, N > 1, started by hand: become the launcher.  This process
    never initialises HIP (no torch.cuda call, no libbfhip call); it starts N fresh rank
    processes with the torch.distributed environment, relays rank 0's JSON line and returns
    the ranks' return code -- the reference's host likewise forks its own filter processes
    (bfrun.c:2312-2328).  A rank that fails takes the others down (by PID)."""
    import signal
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BFHIP_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # this pool's driver only does dmabuf IPC
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True,
                                      start_new_session=True, preexec_fn=_die_with_parent))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("BFHIP_BENCH_TIMEOUT", "1500"))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is not None:
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with %d\n" % (r, code))
        if (rc != 0 or time.time() > deadline) and alive:
            if rc == 0:
                rc = 124
                sys.stderr.write("bench.py: ranks still running at the time limit\n")
            for r in alive:                                  # exactly the processes started above
                try:
                    os.killpg(procs[r].pid, signal.SIGKILL)
                except OSError:
                    pass
            for r in alive:
                procs[r].wait()
            alive = set()
        if alive:
            time.sleep(0.05)
    reader.join(timeout=10)
    lines = [ln for ln in out0 if ln.startswith("{")]
    if rc == 0 and not lines:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        rc = 1
    if lines:                         # also when a rank failed afterwards (verify.ok false): the line says why
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--workload", default="C", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="input", choices=["input", "output"],
                    help="how --gpus N > 1 split the crossbar: input = north_star's partitioning (rank r owns "
                         "I/N inputs, one RCCL reduce-scatter of the partial output spectra per block); "
                         "output = the reference's own process rule (bfconf.c:2893-2931: rank r owns O/N outputs "
                         "and every filter feeding them, transforms ALL inputs itself, no collective at all)")
    ap.add_argument("--pairs", action="store_true",
                    help="INFORMATIVE, never the headline: two blocks per pass over the coefficients "
                         "(bfhip_engine_block_pair_dev) -- what a host with two periods in hand gets; one period "
                         "of extra I/O delay, the same output bits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the end-to-end check of the timed configuration's outputs (after the timed region)")
    ap.add_argument("--host-io", action="store_true",
                    help="time bfhip_engine_block() with HOST raw buffers (PCIe both ways and a "
                         "sync per block included) -- informative, never the headline value")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help=argparse.SUPPRESS)
    ap.add_argument("--ref-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--ref-shape", default="8192,4", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.ref_worker is not None:
        Lr, rsr = (int(v) for v in args.ref_shape.split(","))
        print(json.dumps(_ref_loop_worker(Lr, rsr, args.cpu_seconds, 128)), flush=True)
        return
    if args.cpu_worker is not None:
        # one core's share of the CPU baseline (a child of cpu_baseline()): oracle only, no GPU
        print(json.dumps(_cpu_worker((WORKLOADS[args.workload], args.cpu_worker, args.cpu_seconds))), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # no outside launcher: start the ranks ourselves, before anything here touches the GPU
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import brutefir_amd as bf
    from brutefir_amd import sharding

    dist = None
    backend = os.environ.get("BFHIP_DIST_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("bench.py: BFHIP_DIST_BACKEND must be nccl (RCCL) or gloo")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible (there is no CPU path)")
    if world > ndev and backend != "gloo":
        # several ranks on one device would measure something else under the RCCL headline (and
        # RCCL refuses duplicate devices); the gloo rehearsal has to be asked for explicitly
        raise SystemExit("bench.py: --gpus %d but only %d device(s) visible; set BFHIP_DIST_BACKEND=gloo "
                         "for a rehearsal with ranks sharing a device" % (world, ndev))
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    ranks_seen = [0]
    # BFHIP_BENCH_REHEARSE_RCCL=1 (with BFHIP_BENCH_REHEARSE_RANKS=N, one GPU): the rehearsal's slice
    # copy becomes a real RCCL reduce-scatter on a ONE-rank communicator -- process-group set-up,
    # device barrier, the asynchronous work handle and RCCL's own stream are then the calls and the
    # ordering of the N > 1 run; only the peers are missing
    rccl_rehearsal = world == 1 and os.environ.get("BFHIP_BENCH_REHEARSE_RCCL") == "1"
    if rccl_rehearsal:
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        backend = "nccl"
    if world > 1 or rccl_rehearsal:
        import torch.distributed as dist
        # an RCCL failure is fatal (non-zero exit): never a different transport under this headline
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        got = [None] * world
        dist.all_gather_object(got, {"rank": rank, "device": dev_index, "pid": os.getpid()})
        ranks_seen = sorted(g["rank"] for g in got)
        if ranks_seen != list(range(world)):
            raise SystemExit("bench.py: ranks seen %s, expected 0..%d" % (ranks_seen, world - 1))
        if backend == "nccl":                        # first device collective = communicator set-up
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise SystemExit("bench.py: RCCL all-reduce returned %s, expected %d" % (probe.item(), world))

    wl = WORKLOADS[args.workload]
    I, O, L, N, rs, fmt = wl
    # Rehearsal of ONE rank of an N-rank run on a single GPU (BFHIP_BENCH_REHEARSE_RANKS=N): the very
    # step loop of the N > 1 path -- phase calls, fused [K3 | K1] launch, three buffers in flight --
    # with the collective replaced by a device copy of this rank's slice.  What a rank costs before
    # any communication; not a bench line (the metric name says so).
    rehearse = int(os.environ.get("BFHIP_BENCH_REHEARSE_RANKS", "0")) if world == 1 else 0
    shards = rehearse if rehearse > 1 else world
    fi, ci, fo, co = sharding.shard_crossbar(I, O, shards, rank)
    # --shard output: this rank's engine sees ALL inputs (it transforms them itself, redundantly: 64
    # transforms are ~20 us beside a MAC of hundreds) and owns outputs [fo, fo+co) with every filter
    # that feeds them: the one-GPU block call, no phase calls, no partial sums leave the engine, no
    # collective.  The engine is the compact one of exactly that share (the host patch's shard-of-the-
    # whole-configuration engine has the same MAC entries for the groups it owns).
    out_sharded = shards > 1 and args.shard == "output"
    if out_sharded:
        fi, ci = 0, I

    eng = bf.Engine(L, N, rs, ci, co if out_sharded else O, device=dev_index)
    infmts = bf.interleaved_formats(fmt, I)
    for c in range(ci):
        eng.set_format(bf.IN, c, infmts[fi + c])
    outfmts = bf.interleaved_formats(fmt, O)
    for c in range(co if out_sharded else O):
        eng.set_format(bf.OUT, c, outfmts[fo + c] if out_sharded else outfmts[c])
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.pairs:
        if shards != 1 or args.host_io or args.steps % 2 or args.warmup % 2:
            raise SystemExit("bench.py: --pairs is a single-GPU device-buffer mode with even --steps / --warmup")
        eng.enable_pairs(True)
    taps = L * N
    tdt = torch.float32 if rs == 4 else torch.float64
    my_outs = range(fo, fo + co) if out_sharded else range(O)
    n_sets = sum(1 for o in my_outs for i in range(fi, fi + ci) if not (args.workload in DIAGONAL and i != o))
    # what bfconf knows before it loads the first file: n_coeffs x n_blocks x cbufsize
    eng.reserve_coeffs(float(n_sets) * N * 2 * L * rs)
    pairs = [(o, i) for o in my_outs for i in range(fi, fi + ci)]
    if os.environ.get("BFHIP_BENCH_ORDER") == "input-major":      # experiment: placement order of the sets
        pairs = [(o, i) for i in range(fi, fi + ci) for o in my_outs]
    for o, i in pairs:
        if args.workload in DIAGONAL and i != o:
            continue
        h = synth_ir_dev(torch, 4321 + o * I + i, taps, 1 if args.workload in DIAGONAL else I, device).to(tdt)
        c = eng.add_coeff_dev(h, taps)
        eng.add_filter(in_ch=[i - fi], out_ch=[o - fo if out_sharded else o], coeff=c)
    torch.cuda.synchronize()
    eng.finalize()
    alg = eng.algorithmic_bytes()
    # rings count as full from the first block on: every launch of the run, warm-up included, does
    # the steady-state work, so rocprofv3 --stats averages agree with the live figure below
    eng.prewarm()

    n_pool = 4
    if fmt == "S24_4LE":
        raw_in = synth_raw_blocks(torch, n_pool, L, I, device)
        raw_out = torch.zeros(L, O, dtype=torch.int32, device=device)
    else:
        g = torch.Generator(device=device)
        g.manual_seed(1234)
        raw_in = (torch.randn(n_pool, L, I, generator=g, device=device, dtype=torch.float64) * 0.1).contiguous()
        raw_out = torch.zeros(L, O, dtype=torch.float64, device=device)
    pipelined = shards > 1 and not out_sharded and not os.environ.get("BFHIP_BENCH_SYNC_COLLECTIVE")

    class _Done:
        """stand-in for a torch Work handle when the collective already ran on the host (gloo)"""
        def wait(self):
            return True

    def start_mixdown(zp, zl):
        if rehearse > 1 and rccl_rehearsal:
            return dist.reduce_scatter_tensor(zl, zp[fo:fo + co], async_op=True)   # one-rank communicator
        if rehearse > 1:
            zl.copy_(zp[fo:fo + co])                 # stands in for the reduce-scatter, on the compute stream
            return _Done()
        if backend == "nccl":
            return dist.reduce_scatter_tensor(zl, zp, async_op=True)      # on RCCL's own stream
        torch.cuda.synchronize()
        sharding.mixdown(zp, zl)
        return _Done()
    depth = 3
    if shards > 1 and not out_sharded:
        z_part = [torch.zeros(O, L, 2, dtype=tdt, device=device) for _ in range(depth)]
        z_loc = [torch.zeros(co, L, 2, dtype=tdt, device=device) for _ in range(depth)]
    pending = []          # (work handle, buffer index) of blocks whose mix-down is in flight
    timed = [False]       # inside the timed region
    wait_events = []      # (before, after) event pairs around the stream's wait for a collective

    host_in = [raw_in[i].cpu().numpy() for i in range(n_pool)] if args.host_io else None
    host_inflight = [0]
    if args.host_io:
        # host buffers the way a patched bfrun hands them over: pinned double buffer, upload of
        # block t+1 and download of block t-1 on the copy engines while block t computes
        import ctypes
        host_out = np.zeros(raw_out.numel() * raw_out.element_size(), np.uint8)
        eng.rt_begin(bf.RT_OVERLAP)
        _lib = bf.lib()
        _out_p = ctypes.c_void_p(host_out.ctypes.data)
        _in_p = [ctypes.c_void_p(a.ctypes.data) for a in host_in]

    # which device buffers block k reads and writes: the pool and the one output buffer while timing,
    # per-block buffers during the verification pass (same step loop, same launches)
    raw_out2 = torch.zeros_like(raw_out) if args.pairs else None
    bufs = {"in": lambda k: raw_in[k % n_pool], "out": lambda k: raw_out if (k % 2 == 0 or not args.pairs) else raw_out2}

    def step(k):
        src = bufs["in"](k)
        if shards == 1 or out_sharded:
            if args.host_io:
                if _lib.bfhip_engine_rt_submit(eng.h, _in_p[k % n_pool]) < 0:
                    raise RuntimeError(_lib.bfhip_last_error().decode())
                host_inflight[0] += 1
                if host_inflight[0] == 2:
                    if _lib.bfhip_engine_rt_wait(eng.h, _out_p, None) < 0:
                        raise RuntimeError(_lib.bfhip_last_error().decode())
                    host_inflight[0] -= 1
            elif args.pairs:
                if k % 2 == 1:                       # blocks k - 1 and k: one pass over the coefficients
                    eng.block_pair_dev(bufs["in"](k - 1), bufs["out"](k - 1), src, bufs["out"](k))
            else:
                eng.block_dev(src, bufs["out"](k))
            return
        b = k % depth
        if pipelined:
            # Two blocks in flight: the RCCL reduce-scatter of block k-1 travels while block k is
            # computed, and the inverse FFTs of block k-2 share ONE launch with the forward FFTs
            # of block k (both are a handful of workgroups).  Everything is inside the timed
            # region; the pipeline is drained before the clock stops.
            if len(pending) == 2:
                work, pb, kb = pending.pop(0)
                if timed[0] and k % 4 == 0:
                    # how long the compute stream stalls for the collective (= what of it is NOT
                    # hidden behind the previous block's kernels)
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record()
                    work.wait()                      # current stream waits for that collective
                    eb.record()
                    wait_events.append((ea, eb))
                else:
                    work.wait()
                eng.outputs_inputs_dev(z_loc[pb], fo, co, bufs["out"](kb), src)
            else:
                eng.inputs_dev(src)
            eng.mac_dev(z_part[b])
            eng.advance()
            pending.append((start_mixdown(z_part[b], z_loc[b]), b, k))
        else:
            eng.inputs_dev(src)
            eng.mac_dev(z_part[b])
            eng.advance()
            sharding.mixdown(z_part[b], z_loc[b])
            eng.outputs_dev(z_loc[b], fo, co, bufs["out"](k))

    status_acc = [0]

    def drain():
        if (shards == 1 or out_sharded) and not args.host_io:
            # the engine may still owe the output passes of the last one or two blocks (deferred /
            # ping-pong schedule): they belong to the steps just issued, so they are flushed -- and
            # waited for -- INSIDE the timed region
            status_acc[0] |= eng.sync()
        while host_inflight[0]:
            if _lib.bfhip_engine_rt_wait(eng.h, _out_p, None) < 0:
                raise RuntimeError(_lib.bfhip_last_error().decode())
            host_inflight[0] -= 1
        while pending:
            work, pb, kb = pending.pop(0)
            work.wait()
            eng.outputs_dev(z_loc[pb], fo, co, bufs["out"](kb))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    drain()
    fence()
    # per-kernel HIP events on every 4th block of the timed region: the records themselves
    # cost the stream ~20 us per block, which would otherwise be charged to `value`
    # (small crossbars are bound by the host's launch rate: there every 16th block only)
    stride = (16 if eng.block_mode in (1, 3) else 4) if args.steps >= 32 else 1
    eng.enable_timing(0 if os.environ.get("BFHIP_BENCH_NO_EVENTS") else stride)
    timed[0] = True
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    drain()
    fence()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    status = eng.sync() | status_acc[0]
    tm = eng.timing()
    src_hash = source_hash()
    eng.enable_timing(0)

    def verify():
        """End-to-end check of the very configuration that was just timed, through the same step
        loop (fused launches, outputs owed for one or two calls, and -- at N > 1 -- the real
        mix-down collective): N silent blocks empty the rings, ONE block of independent noise on
        EVERY input channel follows, then N silent blocks.  Every output of this rank over those
        N + 1 blocks must be the sum over all inputs of the linear convolution of that block with
        the very impulse response that was loaded for the (output, input) pair -- every coefficient
        set of the configuration is exercised (config C: all 4096) -- computed here in float64 with
        torch.fft (the impulse responses are regenerated from their seeds on the device)."""
        timed[0] = False
        k0 = args.warmup + args.steps
        nb = 2 * N + 1 + (1 if args.pairs else 0)       # (pairs: an even number of blocks)
        probe = list(range(I))
        if rehearse > 1 and not out_sharded:
            probe = list(range(fi, fi + ci))    # one input-sharded rank alone: only its own inputs reach its outputs
        x = torch.zeros_like(raw_in[0])
        x[:, probe] = raw_in[0][:, probe]
        silence = torch.zeros_like(raw_in[0])
        vout = [torch.zeros_like(raw_out) for _ in range(nb)]
        bufs["in"] = lambda k: x if k - k0 == N else silence
        bufs["out"] = lambda k: vout[k - k0]
        for k in range(k0, k0 + nb):
            step(k)
        drain()
        st = eng.sync()
        torch.cuda.synchronize()
        got = torch.cat(vout[N:2 * N + 1], dim=0)[:, fo:fo + co].to(torch.float64)       # [(N+1) L][co]
        n_y = (N + 1) * L
        n_fft = 1
        while n_fft < L + taps:
            n_fft *= 2
        integer_io = fmt == "S24_4LE"
        xs = x[:, probe].to(torch.float64)                                      # counts (S24) or reals
        X = torch.fft.rfft(xs, n=n_fft, dim=0)                                  # [n_fft/2+1][probe]
        # (BFHIP_BENCH_VERIFY_SELFTEST=1, tests only: expect the WRONG filter behind ONE interior
        # (output, input) pair -- the check has to notice and the run has to fail)
        selftest = os.environ.get("BFHIP_BENCH_VERIFY_SELFTEST") == "1"
        wrong_pair = (fo + co // 2, probe[len(probe) // 2])
        if args.workload in DIAGONAL:
            wrong_pair = (fo + co // 2, fo + co // 2)
        worst, worst_rel, checked, sets, peak = 0.0, 0.0, 0, 0, 0.0
        for o in range(fo, fo + co):
            W = torch.zeros(n_fft // 2 + 1, dtype=torch.complex128, device=device)
            for j, i in enumerate(probe):
                if args.workload in DIAGONAL and i != o:
                    continue
                wrong = 1 if (selftest and (o, i) == wrong_pair) else 0
                h = synth_ir_dev(torch, 4321 + o * I + i + wrong, taps, 1 if args.workload in DIAGONAL else I, device).to(tdt)
                W += X[:, j] * torch.fft.rfft(h.to(torch.float64), n=n_fft)
                sets += 1
            want = torch.fft.irfft(W, n=n_fft)[:n_y]
            err = (got[:, o - fo] - want).abs().max().item()
            worst = max(worst, err)
            peak = max(peak, want.abs().max().item())
            worst_rel = max(worst_rel, err / max(want.abs().max().item(), 1e-300)) if want.abs().max().item() > 0 else worst_rel
            checked += 1
        # integer output: see INTEGER_VERIFY_BAR; float64 in and out: the working precision
        bar = (INTEGER_VERIFY_BAR[0] + max(INTEGER_VERIFY_BAR[1], INTEGER_VERIFY_BAR[2] * peak)) if integer_io else 1e-9
        ok = (worst <= bar) if integer_io else (worst_rel <= bar)
        return {"ok": bool(ok and st == 0), "status_bits": st, "outputs_checked": checked, "blocks": N + 1,
                "probe_inputs": probe, "coefficient_sets_exercised": sets,
                "max_abs_err": worst, "max_err_rel_to_peak": worst_rel, "bar": bar, "output_peak": peak,
                "unit": "LSB of S24" if integer_io else "output units (bar: relative to the peak)",
                "against": "float64 torch.fft: per output, the sum over all probe inputs of the convolution of the "
                           "noise block with the impulse response loaded for that (output, input) pair"}

    ver = None
    if not args.no_verify and not args.host_io:
        ver = verify()
        if dist is not None:
            allv = [None] * world
            dist.all_gather_object(allv, ver)
            ver = dict(allv[0], ok=all(v["ok"] for v in allv), max_abs_err=max(v["max_abs_err"] for v in allv),
                       max_err_rel_to_peak=max(v["max_err_rel_to_peak"] for v in allv),
                       outputs_checked=sum(v["outputs_checked"] for v in allv),
                       coefficient_sets_exercised=sum(v["coefficient_sets_exercised"] for v in allv), ranks=world)

    def mac_roofline(tm_, alg_mac):
        mac_s = tm_["mac_ms"] * 1e-3
        ach = alg_mac / mac_s / 1e9 if mac_s > 0 else None
        return {"bound": "hbm", "kernel": "mac_diag_kernel" if eng.uses_diag_mac else "mac_xbar_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS if ach else None,
                "algorithmic_bytes_per_launch": alg_mac, "avg_launch_ms": tm_["mac_ms"],
                "timed_launches": tm_["launches"]}

    per_rank = None
    if dist is not None:
        exposed = [a.elapsed_time(b) for a, b in wait_events]
        mine = {"rank": rank, "device": dev_index, "status_bits": status,
                "inputs": [fi, fi + ci], "outputs": [fo, fo + co], "shard": args.shard,
                "mac_ms": tm["mac_ms"], "io_ms": tm["fft_in_ms"],
                "exposed_collective_ms": sum(exposed) / len(exposed) if exposed else None,
                "roofline": mac_roofline(tm, alg["mac"])}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        ms = el * 1e3 / args.steps
        value = (co if rehearse > 1 else O) * L * args.steps / el      # rehearsal: this rank's outputs only
        out = {
            "metric": ("REHEARSAL of one rank of %d (%s): rank-local samples/sec"
                       % (rehearse, "output-sharded: no collective exists" if out_sharded
                          else "RCCL reduce-scatter on a one-rank communicator" if rccl_rehearsal
                          else "input-sharded, collective replaced by a copy")) if rehearse > 1
                      else ("INFORMATIVE (not the headline): block pairs -- two blocks per pass over the coefficients, "
                            "one period of extra I/O delay: filtered samples/sec") if args.pairs
                      else (BASELINE_METRIC if args.workload == "C" else "filtered samples/sec"),
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32" if rs == 4 else "f64",
            "data": "synthetic (seeded noise PCM, seeded decaying-noise IRs)",
            "config": {"workload": "%d-in/%d-out %s, %d taps (%d x %d partitions), "
                                   "%s, %d filters" % (I, O, "one-to-one" if args.workload in DIAGONAL
                                                       else "full crossbar", L * N, L, N, fmt,
                                                       I if args.workload in DIAGONAL else I * O),
                       "baseline_config": {"C": "configs[2]", "B": "configs[1]", "D": "configs[3] on one GPU",
                                           "E": "configs[4] (uniform partitions)"}.get(args.workload, "per-rank share of configs[2]"),
                       "parallelism": ("output-sharded x%d: every rank transforms all inputs, owns %d outputs and their "
                                       "filters; no data-path collective" % (world, co)) if (world > 1 and out_sharded)
                                      else ("input-sharded x%d + %s reduce-scatter%s"
                                            % (world, "RCCL" if backend == "nccl" else "gloo (host, rehearsal)",
                                               " (overlapped with the next block)" if pipelined else ""))
                                      if world > 1
                                      else "single GPU",
                       "status_bits": status if per_rank is None else max(p["status_bits"] for p in per_rank),
                       "block_schedule": {0: "K1, MAC, K3 in order on one stream",
                                          1: "pipelined: K1 of t+1 and K3 of t-1 on side streams beside the MAC of t",
                                          2: "deferred output: [K3 of t-1 | K1 of t] in one launch, then the MAC of t",
                                          3: "ping-pong: [K3 of t-2 | K1 of t] on a side stream beside the MAC of t-1"
                                          }.get(eng.block_mode, "?") if (shards == 1 or out_sharded) else "phase calls + fused [K3 | K1] launch",
                       "fft": "wave-level (fft_wave.h)" if eng.uses_wave_fft else "LDS Stockham (fft_lds.h)"},
            "hbm_gbs_algorithmic": alg["block"] / (ms * 1e-3) / 1e9 if world == 1 else None,
        }
        if args.host_io:
            out["config"]["io"] = "host buffers through bfhip_engine_rt_submit/rt_wait, two blocks in flight (PCIe-inclusive)"
        out["source_hash"] = src_hash
        out["verify"] = ver
        if world == 1:
            traffic, traffic_stale, traffic_src = None, None, None
            tp = os.path.join(ROOT, "profiles", "traffic_config%s.json" % args.workload)
            if os.path.exists(tp) and rehearse <= 1:
                # HBM bytes per MAC launch from the committed rocprofv3 PMC passes of this very
                # workload (counters cannot be read from inside this process).  The file names
                # the sources it was measured on; a different binary is flagged, not trusted.
                tj = json.load(open(tp))
                traffic = tj["traffic_bytes_per_launch"]
                traffic_src = tj.get("source_hash")
                traffic_stale = traffic_src != src_hash
            rf = mac_roofline(tm, alg["mac"])
            if args.pairs:
                # one paired launch: the coefficients and the rings once, one more ring slot per input,
                # a second set of output spectra
                Cb = L * 2 * rs
                pair_bytes = alg["mac"] + Cb * (ci + O)
                mac_s = tm["mac_ms"] * 1e-3
                rf.update({"kernel": "mac_xbar2_kernel", "algorithmic_bytes_per_launch": pair_bytes,
                           "achieved": pair_bytes / mac_s / 1e9 if mac_s > 0 else None,
                           "frac": pair_bytes / mac_s / 1e9 / HBM_PEAK_GBS if mac_s > 0 else None,
                           "blocks_per_launch": 2, "paired_launches": eng.pair_launches})
                traffic = None
            rf.update({"traffic": traffic, "traffic_stale": traffic_stale, "traffic_source_hash": traffic_src,
                       "launches": args.steps, "fft_in_ms": tm["fft_in_ms"], "ifft_out_ms": tm["ifft_out_ms"]})
            if eng.block_mode in (2, 3):
                rf["fft_note"] = "fft_in_ms is the fused [K3 of t-1 | K1 of t] launch; there is no separate K3 launch"
            out["roofline"] = rf
            if rccl_rehearsal:
                out["exposed_collective_ms"] = per_rank[0]["exposed_collective_ms"]
                out["backend"] = "rccl (one-rank communicator)"
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(wl)
        else:
            # the dominant kernel on the slowest rank prices the job; every rank's own figure beside it
            slow = max(per_rank, key=lambda p_: p_["mac_ms"])
            rf = dict(slow["roofline"])
            rf.update({"traffic": None, "rank": slow["rank"],
                       "note": "per-rank MAC launch (%s); slowest rank shown"
                               % (("all inputs, 1/%d of the outputs" if out_sharded else "1/%d of the inputs, all outputs") % world)})
            out["roofline"] = rf
            out["ranks_seen"] = ranks_seen
            out["per_rank"] = per_rank
            ex = [p_["exposed_collective_ms"] for p_ in per_rank if p_["exposed_collective_ms"] is not None]
            out["exposed_collective_ms"] = max(ex) if ex else None
            out["backend"] = "rccl" if backend == "nccl" else "gloo"
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if ver is not None and not ver["ok"]:
        raise SystemExit("bench.py: the outputs of the timed configuration are WRONG: %s" % json.dumps(ver))


if __name__ == "__main__":
    main()
