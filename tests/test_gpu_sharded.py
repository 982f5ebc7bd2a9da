"""GPU rehearsal of the multi-GPU decomposition on ONE card: two ranks (gloo, both on cuda:0)
each run the HIP engine on half of the inputs, exchange partial output spectra through
brutefir_amd.sharding.mixdown and inverse-transform their half of the outputs; the result must
equal one engine running the whole crossbar.  (RCCL itself cannot be exercised with two ranks
on one device; the 8-GPU run is the driver's.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

L, N, I, O = 1024, 4, 4, 4
FMT = "S24_4LE"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    import brutefir_amd as bf
    from brutefir_amd import sharding
    import cases
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    fi, ci, fo, co = sharding.shard_crossbar(I, O, world, rank)
    e = bf.Engine(L, N, 4, ci, O)
    fm = bf.interleaved_formats(FMT, I)
    for c in range(ci):
        e.set_format(bf.IN, c, fm[fi + c])
    for c, f in enumerate(bf.interleaved_formats("FLOAT_LE", O)):
        e.set_format(bf.OUT, c, f)
    for o in range(O):
        for i in range(fi, fi + ci):
            h = cases.make_ir(np.random.default_rng(4321 + o * I + i), L * N, I)
            e.add_filter(in_ch=[i - fi], out_ch=[o], coeff=e.add_coeff(h))
    e.finalize()
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    z_part = torch.zeros(O, L, 2, dtype=torch.float32, device=dev)
    z_loc = torch.zeros(co, L, 2, dtype=torch.float32, device=dev)
    raw_out = torch.zeros(L, O, dtype=torch.float32, device=dev)
    outs = []
    for blk in cases.raw_blocks(1234, N + 3, L, I, FMT):
        src = torch.from_numpy(blk).to(dev)
        e.inputs_dev(src)
        e.mac_dev(z_part)
        torch.cuda.synchronize()
        sharding.mixdown(z_part, z_loc)
        e.outputs_dev(z_loc, fo, co, raw_out)
        e.advance()
        torch.cuda.synchronize()
        outs.append(raw_out[:, fo:fo + co].cpu().numpy().copy())
    st = e.sync()
    dist.destroy_process_group()
    q.put((rank, fo, co, st, np.stack(outs)))


def test_two_ranks_on_one_gpu_match_the_unsharded_engine(hip):
    import torch.multiprocessing as mp
    import cases
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full, _ = cases.crossbar(hip.Engine, L, N, 4, I, O, FMT, "FLOAT_LE")
    want = []
    for blk in cases.raw_blocks(1234, N + 3, L, I, FMT):
        st, raw = full.block(blk)
        assert st == 0
        want.append(np.frombuffer(raw.tobytes(), np.float32).reshape(L, O).copy())
    want = np.stack(want)
    for rank, fo, co, st, got in res:
        assert st == 0
        err = cases.rel_rms(got, want[:, :, fo:fo + co])
        assert err <= 1e-6, (rank, err)        # same kernels, only the summation order differs


def test_phase_api_and_fused_io_launch_equal_the_block_call(hip):
    """inputs / mac / outputs as separate phases, and the fused `outputs of block k-2 + inputs
    of block k` launch bench.py uses at N > 1, give bit-identical samples to block()"""
    import torch
    import cases
    Lq, Nq, Iq, Oq = 2048, 4, 3, 5
    dev = torch.device("cuda", 0)
    ref_e, _ = cases.crossbar(hip.Engine, Lq, Nq, 4, Iq, Oq, FMT, "FLOAT_LE")
    pipe_e, _ = cases.crossbar(hip.Engine, Lq, Nq, 4, Iq, Oq, FMT, "FLOAT_LE")
    pipe_e.set_stream(torch.cuda.current_stream().cuda_stream)
    blocks = cases.raw_blocks(77, Nq + 5, Lq, Iq, FMT)
    want = []
    for blk in blocks:
        st, raw = ref_e.block(blk)
        assert st == 0
        want.append(np.frombuffer(raw.tobytes(), np.float32).reshape(Lq, Oq).copy())
    srcs = [torch.from_numpy(b).to(dev) for b in blocks]
    z = [torch.zeros(Oq, Lq, 2, dtype=torch.float32, device=dev) for _ in range(3)]
    out = torch.zeros(Lq, Oq, dtype=torch.float32, device=dev)
    got = {}
    for k, src in enumerate(srcs):
        if k >= 2:
            pipe_e.outputs_inputs_dev(z[(k - 2) % 3], 0, Oq, out, src)
            torch.cuda.synchronize()
            got[k - 2] = out.cpu().numpy().copy()
        else:
            pipe_e.inputs_dev(src)
        pipe_e.mac_dev(z[k % 3])
        pipe_e.advance()
    for k in (len(srcs) - 2, len(srcs) - 1):
        pipe_e.outputs_dev(z[k % 3], 0, Oq, out)
        torch.cuda.synchronize()
        got[k] = out.cpu().numpy().copy()
    assert pipe_e.sync() == 0
    for k in range(len(srcs)):
        assert np.array_equal(got[k], want[k]), k


def test_bench_two_rank_pipeline_on_one_gpu():
    """bench.py's N > 1 code path end to end, started the way a user starts it: plain
    `python bench.py --gpus 2`, no outside launcher.  The parent forks its own two ranks (like
    the reference's host forks its filter processes, bfrun.c:2312-2328); they share cuda:0 over
    gloo (asked for explicitly -- without BFHIP_DIST_BACKEND=gloo two ranks on one device are an
    error, next test).  The JSON line must come out, name both ranks and report no status bits."""
    import json
    import subprocess
    env = dict(os.environ, BFHIP_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8",
                        "--warmup", "10", "--workload", "B"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["value"] > 0
    assert d["config"]["status_bits"] == 0 and d["scaling"] == "strong"
    assert d["ranks_seen"] == [0, 1] and d["backend"] == "gloo"
    assert [p["rank"] for p in d["per_rank"]] == [0, 1]
    for p in d["per_rank"]:
        assert p["mac_ms"] > 0 and p["roofline"]["achieved"] > 0
        assert p["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert d["roofline"]["frac"] > 0 and "exposed_collective_ms" in d
    # the line carries its own end-to-end check of the sharded outputs (mix-down included)
    v = d["verify"]
    assert v["ok"] and v["ranks"] == 2 and v["outputs_checked"] == 8 and v["max_abs_err"] <= v["bar"]
    assert v["coefficient_sets_exercised"] == 64


def test_bench_output_sharded_two_ranks_on_one_gpu():
    """`--shard output`: the reference's own process rule (every output mixed inside one process,
    bfconf.c:2893-2931) as the multi-GPU split -- rank r transforms ALL inputs itself, owns O/N
    outputs and their filters, and there is no data-path collective at all (the rendezvous, the
    barriers around the timed region and the max-over-ranks time are all that is exchanged)."""
    import json
    import subprocess
    env = dict(os.environ, BFHIP_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shard", "output", "--steps", "8",
                        "--warmup", "10", "--workload", "B"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["status_bits"] == 0
    assert "output-sharded" in d["config"]["parallelism"] and "no data-path collective" in d["config"]["parallelism"]
    assert [(p["inputs"], p["outputs"], p["shard"]) for p in d["per_rank"]] == [([0, 8], [0, 4], "output"), ([0, 8], [4, 8], "output")]
    assert d["exposed_collective_ms"] is None
    v = d["verify"]
    assert v["ok"] and v["ranks"] == 2 and v["outputs_checked"] == 8 and v["coefficient_sets_exercised"] == 64


def test_bench_refuses_two_ranks_on_one_device_under_the_rccl_headline():
    """no silent change of transport or topology: `--gpus 2` on a one-GPU box without the
    explicit gloo switch exits non-zero and prints no result line"""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with a single GPU")
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BFHIP_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "2", "--workload", "B"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "BFHIP_DIST_BACKEND=gloo" in r.stderr


def test_bench_step_loop_with_a_real_rccl_reduce_scatter_on_one_rank():
    """The RCCL calls of the N > 1 step loop -- process-group set-up, device barrier, asynchronous
    `reduce_scatter_tensor` on RCCL's own stream, the work handle the compute stream waits on, the
    object all-gather -- executed on THIS device with a one-rank communicator
    (BFHIP_BENCH_REHEARSE_RCCL=1: rank 0 of 4 reduce-scatters its own slice).  Only the peers are
    missing; what runs on the driver's 8-GPU node is this code."""
    import json
    import subprocess
    env = dict(os.environ, BFHIP_BENCH_REHEARSE_RANKS="4", BFHIP_BENCH_REHEARSE_RCCL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BFHIP_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "6",
                        "--workload", "B", "--no-cpu-baseline"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["backend"].startswith("rccl") and "RCCL reduce-scatter" in d["metric"]
    assert d["config"]["status_bits"] == 0 and d["value"] > 0
    assert d["exposed_collective_ms"] is not None and d["exposed_collective_ms"] >= 0
    assert d["verify"]["ok"] and d["verify"]["max_abs_err"] <= d["verify"]["bar"]      # outputs behind the RCCL call are right


def test_bench_verification_notices_wrong_outputs():
    """bench.py checks the outputs of the configuration it has just timed against a float64
    convolution with the loaded impulse responses and fails the run when they differ.  With the
    test hook that makes it EXPECT another filter the run must exit non-zero and say so (the line
    is still printed, `verify.ok` false); without the hook the same run passes."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BFHIP_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "4", "--workload", "B",
           "--no-cpu-baseline"]
    good = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert good.returncode == 0, good.stderr[-2000:]
    v = json.loads([ln for ln in good.stdout.splitlines() if ln.startswith("{")][0])["verify"]
    # noise on EVERY input, every output checked: all 64 coefficient sets of workload B are behind the
    # result; an integer output must be the exact value rounded to the nearest count
    assert v["ok"] and v["outputs_checked"] == 8 and v["probe_inputs"] == list(range(8))
    assert v["coefficient_sets_exercised"] == 64 and 0.501 <= v["bar"] < 0.52 and 0.4 < v["max_abs_err"] <= v["bar"]
    # ONE interior (output, input) pair with the wrong filter: the check has to notice
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=dict(env, BFHIP_BENCH_VERIFY_SELFTEST="1"))
    assert bad.returncode != 0 and "WRONG" in bad.stderr
    v = json.loads([ln for ln in bad.stdout.splitlines() if ln.startswith("{")][0])["verify"]
    assert not v["ok"] and v["max_abs_err"] > 10.0
