#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one block of the hot path over one batch of synthetic input: L new samples on
each of the 64 input channels -> forward FFTs -> 64x64 crossbar multiply-accumulate over 32
coefficient partitions per filter (4096 independent 262144-tap IRs, 8 GiB of coefficients in
HBM) -> inverse FFTs -> 64 x L output samples.  Inputs are resident in HBM when the timed region
starts.  N > 1: the crossbar is sharded by input channel (brutefir_amd/sharding.py), one
process per GPU, one RCCL reduce-scatter per block; total work is fixed ("strong" scaling).

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
    # a float64 crossbar of the headline's byte volume per filter (informative: the f64 MAC path)
    "F": (32, 32, 8192, 32, 8, "FLOAT64_LE"),
}
DIAGONAL = {"D"}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
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
    raw = (rng.standard_normal((L, I)) * 0.1 * 8388608).astype(np.int32)
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
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16, max(1, O // 2)))
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
    ref_loop = reference_hot_loop(bo, L, rs, I, N)
    return {"value": agg, "unit": "samples/s", "cores": cores, "kind": "port",
            "single_core_value": single[0],
            "reference_hot_loop": ref_loop,
            "sample": "per core: %d of %d outputs x %d inputs x %d partitions of %d taps, %d-%d steady-state "
                      "blocks in %.0f s (input FFTs included), gcc -O2, %d worker processes"
                      % (min(O, 2), O, I, N, L, min(p[1] for p in per), max(p[1] for p in per), seconds, cores)}


def reference_hot_loop(bo, L, rs, I, N, seconds=3.0):
    """The reference's OWN inner loop for this path -- convolve_add, the C version its dispatch
    really selects and the SSE version it intends -- compiled from the reference
    sources into oracle/_ref and timed here on cbufs of the workload's size, cycling over a
    working set larger than the last-level cache.  The full reference binary cannot be built
    (FFTW3 absent); at this workload the loop is > 95 % of the reference's CPU time."""
    R = bo.ref()
    if R is None:
        return None
    R.ref_set_length(L, 0.0)
    dt = np.float32 if rs == 4 else np.float64
    n_bufs = max(4, (512 << 20) // (2 * L * rs * 2))          # ~512 MiB of (b, c) pairs
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
        el = time.time() - t0
        per_call = el / calls
        # one output sample block (L samples) of one output costs I*N such calls
        out[name] = {"us_per_call": per_call * 1e6, "GB_per_s_3_streams": 3 * 2 * L * rs / per_call / 1e9,
                     "equivalent_samples_per_s": L / (I * N * per_call)}
    out["note"] = "1 core, gcc -O2 -msse -msse2, %d-byte cbufs; equivalent rate = MAC only, FFTs excluded" % (2 * L * rs)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--workload", default="C", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-io", action="store_true",
                    help="time bfhip_engine_block() with HOST raw buffers (PCIe both ways and a "
                         "sync per block included) -- informative, never the headline value")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker is not None:
        # one core's share of the CPU baseline (a child of cpu_baseline()): oracle only, no GPU
        print(json.dumps(_cpu_worker((WORKLOADS[args.workload], args.cpu_worker, args.cpu_seconds))), flush=True)
        return

    import torch
    import brutefir_amd as bf
    from brutefir_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with that many ranks" % args.gpus)
    dist = None
    backend = os.environ.get("BFHIP_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    fallback_note = ""
    if world > 1:
        import torch.distributed as dist
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            if backend == "nccl":                    # first collective = communicator set-up
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
        except Exception as exc:                     # noqa: BLE001
            if backend != "nccl":
                raise
            # never silently: the line says which transport carried the mix-down
            sys.stderr.write("bench.py: RCCL not usable here (%s); mix-down over gloo host buffers instead\n" % exc)
            try:
                dist.destroy_process_group()
            except Exception:                        # noqa: BLE001
                pass
            backend = "gloo"
            fallback_note = " [RCCL failed to initialise: gloo host mix-down]"
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    wl = WORKLOADS[args.workload]
    I, O, L, N, rs, fmt = wl
    fi, ci, fo, co = sharding.shard_crossbar(I, O, world, rank)

    eng = bf.Engine(L, N, rs, ci, O, device=dev_index)
    infmts = bf.interleaved_formats(fmt, I)
    for c in range(ci):
        eng.set_format(bf.IN, c, infmts[fi + c])
    for c, f in enumerate(bf.interleaved_formats(fmt, O)):
        eng.set_format(bf.OUT, c, f)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    taps = L * N
    tdt = torch.float32 if rs == 4 else torch.float64
    for o in range(O):
        for i in range(fi, fi + ci):
            if args.workload in DIAGONAL and i != o:
                continue
            h = synth_ir_dev(torch, 4321 + o * I + i, taps, 1 if args.workload in DIAGONAL else I, device).to(tdt)
            c = eng.add_coeff_dev(h, taps)
            eng.add_filter(in_ch=[i - fi], out_ch=[o], coeff=c)
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
    pipelined = world > 1 and not os.environ.get("BFHIP_BENCH_SYNC_COLLECTIVE")

    class _Done:
        """stand-in for a torch Work handle when the collective already ran on the host (gloo)"""
        def wait(self):
            return True

    def start_mixdown(zp, zl):
        if backend == "nccl":
            return dist.reduce_scatter_tensor(zl, zp, async_op=True)      # on RCCL's own stream
        torch.cuda.synchronize()
        sharding.mixdown(zp, zl)
        return _Done()
    depth = 3
    if world > 1:
        z_part = [torch.zeros(O, L, 2, dtype=tdt, device=device) for _ in range(depth)]
        z_loc = [torch.zeros(co, L, 2, dtype=tdt, device=device) for _ in range(depth)]
    pending = []          # (work handle, buffer index) of blocks whose mix-down is in flight

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

    def step(k):
        src = raw_in[k % n_pool]
        if world == 1:
            if args.host_io:
                if _lib.bfhip_engine_rt_submit(eng.h, _in_p[k % n_pool]) < 0:
                    raise RuntimeError(_lib.bfhip_last_error().decode())
                host_inflight[0] += 1
                if host_inflight[0] == 2:
                    if _lib.bfhip_engine_rt_wait(eng.h, _out_p, None) < 0:
                        raise RuntimeError(_lib.bfhip_last_error().decode())
                    host_inflight[0] -= 1
            else:
                eng.block_dev(src, raw_out)
            return
        b = k % depth
        if pipelined:
            # Two blocks in flight: the RCCL reduce-scatter of block k-1 travels while block k is
            # computed, and the inverse FFTs of block k-2 share ONE launch with the forward FFTs
            # of block k (both are a handful of workgroups).  Everything is inside the timed
            # region; the pipeline is drained before the clock stops.
            if len(pending) == 2:
                work, pb = pending.pop(0)
                work.wait()                          # current stream waits for that collective
                eng.outputs_inputs_dev(z_loc[pb], fo, co, raw_out, src)
            else:
                eng.inputs_dev(src)
            eng.mac_dev(z_part[b])
            eng.advance()
            pending.append((start_mixdown(z_part[b], z_loc[b]), b))
        else:
            eng.inputs_dev(src)
            eng.mac_dev(z_part[b])
            eng.advance()
            sharding.mixdown(z_part[b], z_loc[b])
            eng.outputs_dev(z_loc[b], fo, co, raw_out)

    def drain():
        while host_inflight[0]:
            if _lib.bfhip_engine_rt_wait(eng.h, _out_p, None) < 0:
                raise RuntimeError(_lib.bfhip_last_error().decode())
            host_inflight[0] -= 1
        while pending:
            work, pb = pending.pop(0)
            work.wait()
            eng.outputs_dev(z_loc[pb], fo, co, raw_out)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(device_ids=[dev_index]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    drain()
    fence()
    if world == 1:
        # per-kernel HIP events on every 4th block of the timed region: the records themselves
        # cost the stream ~20 us per block, which would otherwise be charged to `value`
        eng.enable_timing(0 if os.environ.get("BFHIP_BENCH_NO_EVENTS") else (4 if args.steps >= 32 else 1))
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
    status = eng.sync()

    if rank == 0:
        ms = el * 1e3 / args.steps
        value = O * L * args.steps / el
        out = {
            "metric": BASELINE_METRIC if args.workload == "C" else "filtered samples/sec",
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
                       "parallelism": ("input-sharded x%d + %s reduce-scatter%s%s"
                                       % (world, "RCCL" if backend == "nccl" else "gloo (host)",
                                          " (overlapped with the next block)" if pipelined else "", fallback_note))
                                      if world > 1
                                      else "single GPU",
                       "status_bits": status},
            "hbm_gbs_algorithmic": alg["block"] / (ms * 1e-3) / 1e9 if world == 1 else None,
        }
        if args.host_io:
            out["config"]["io"] = "host buffers through bfhip_engine_rt_submit/rt_wait, two blocks in flight (PCIe-inclusive)"
        if world == 1:
            tm = eng.timing()
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic_config%s.json" % args.workload)
            if os.path.exists(tp):
                # HBM bytes per MAC launch from the committed rocprofv3 PMC passes of this very
                # workload (counters cannot be read from inside this process)
                traffic = json.load(open(tp))["traffic_bytes_per_launch"]
            mac_s = tm["mac_ms"] * 1e-3
            ach = alg["mac"] / mac_s / 1e9 if mac_s > 0 else None
            out["roofline"] = {"bound": "hbm", "kernel": "mac_xbar_kernel", "achieved": ach,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS if ach else None, "traffic": traffic,
                               "algorithmic_bytes_per_launch": alg["mac"],
                               "avg_launch_ms": tm["mac_ms"], "launches": args.steps, "timed_launches": tm["launches"],
                               "fft_in_ms": tm["fft_in_ms"], "ifft_out_ms": tm["ifft_out_ms"]}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
