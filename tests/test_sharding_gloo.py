"""CPU, world_size 2 and 4 over gloo: the multi-GPU decomposition (brutefir_amd/sharding.py).
Each rank owns its share of the inputs and computes partial output spectra for ALL outputs; one
mix-down collective leaves each rank the finished spectra of its share of the outputs.  The
partial spectra are produced by the oracle here (the HIP engine needs a GPU); what is under
test is the sharding arithmetic and the collective plumbing bench.py uses."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


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
    import bforacle as bo
    import cases
    from brutefir_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, N, I, O = 64, 4, 4, 4
    fi, ci, fo, co = sharding.shard_crossbar(I, O, world, rank)
    # this rank's engine: its inputs only, all outputs; same IR seeds as the full crossbar
    e = bo.Engine(L, N, 8, ci, O)
    fm = bo.interleaved_formats("FLOAT64_LE", I)
    for c in range(ci):
        e.set_format(0, c, fm[fi + c])
    for c, f in enumerate(bo.interleaved_formats("FLOAT64_LE", O)):
        e.set_format(1, c, f)
    for o in range(O):
        for i in range(fi, fi + ci):
            h = cases.make_ir(np.random.default_rng(4321 + o * I + i), L * N, I)
            e.add_filter(in_ch=[i - fi], out_ch=[o], coeff=e.add_coeff(h))
    full, _ = cases.crossbar(bo.Engine, L, N, 8, I, O, "FLOAT64_LE", "FLOAT64_LE")
    worst = 0.0
    for blk in cases.raw_blocks(11, N + 2, L, I, "FLOAT64_LE"):
        e.block(blk)
        full.block(blk)
        zp = np.stack([e.output_spectrum(o) for o in range(O)])           # partial, all outputs
        z_part = torch.from_numpy(zp.copy())
        z_loc = torch.zeros(co, 2 * L, dtype=torch.float64)
        sharding.mixdown(z_part, z_loc)
        want = np.stack([full.output_spectrum(o) for o in range(fo, fo + co)])
        worst = max(worst, float(np.abs(z_loc.numpy() - want).max() / np.abs(want).max()))
    dist.destroy_process_group()
    q.put((rank, worst))


@pytest.mark.parametrize("world", [2, 4])
def test_input_sharded_crossbar_mixdown(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst in res:
        assert worst < 1e-12, (rank, worst)


def _worker_out(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import bforacle as bo
    import cases
    from brutefir_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L, N, I, O = 64, 4, 4, 8
    _, _, fo, co = sharding.shard_crossbar(I, O, world, rank)
    # this rank's engine: ALL inputs (it transforms them itself), its own outputs and every filter
    # that feeds them; its samples go to its own columns of the interleaved output frame
    e = bo.Engine(L, N, 8, I, co)
    e.set_interleaved(0, "FLOAT64_LE")
    fm = bo.interleaved_formats("FLOAT64_LE", O)
    for c in range(co):
        e.set_format(1, c, fm[fo + c])
    e.out_bytes = O * L * 8
    for o in range(fo, fo + co):
        for i in range(I):
            h = cases.make_ir(np.random.default_rng(4321 + o * I + i), L * N, I)
            e.add_filter(in_ch=[i], out_ch=[o - fo], coeff=e.add_coeff(h))
    full, _ = cases.crossbar(bo.Engine, L, N, 8, I, O, "FLOAT64_LE", "FLOAT64_LE")
    exact = True
    for blk in cases.raw_blocks(11, N + 2, L, I, "FLOAT64_LE"):
        _, mine = e.block(blk)                        # no exchange of any kind inside the block
        _, want = full.block(blk)
        frame = torch.from_numpy(mine.view(np.float64).reshape(L, O).copy())
        frames = [torch.zeros_like(frame) for _ in range(world)]
        dist.all_gather(frames, frame)               # (the check only: "the shared output buffer")
        whole = np.zeros((L, O))
        for r, fr in enumerate(frames):
            _, _, fo_r, co_r = sharding.shard_crossbar(I, O, world, r)
            whole[:, fo_r:fo_r + co_r] = fr.numpy()[:, fo_r:fo_r + co_r]
        exact = exact and np.array_equal(whole, want.view(np.float64).reshape(L, O))
    dist.destroy_process_group()
    q.put((rank, exact))


@pytest.mark.parametrize("world", [2, 4])
def test_output_sharded_crossbar_needs_no_collective_and_is_bit_exact(world):
    """the reference's own process rule as the multi-GPU split (bench.py --shard output, the host
    patch with n_processes > 1): every rank transforms all inputs, owns O/world outputs and the
    filters that feed them.  Nothing is exchanged inside a block, and because every output is summed
    by ONE rank in the order a single engine uses, the assembled frame equals the unsharded
    engine's bit for bit (SURVEY B.5 iii observed the same on the reference's processes)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_out, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(exact for _, exact in res), res


def test_shard_arithmetic():
    from brutefir_amd import sharding
    assert sharding.shard_crossbar(64, 64, 8, 3) == (24, 8, 24, 8)
    assert sharding.shard_crossbar(64, 64, 1, 0) == (0, 64, 0, 64)
    with pytest.raises(ValueError):
        sharding.shard_crossbar(26, 26, 8, 0)


def test_load_balance_follows_the_reference_rules():
    """bfconf.c:2227-2318 on the shipped configurations"""
    from brutefir_amd.sharding import load_balance_filters
    # bench1_config: two cascades {0 <- 2, 5} and {1 <- 3, 4} that share inputs only -> two processes
    bench1 = [dict(in_f=[2, 5], out_ch=[0]), dict(in_f=[3, 4], out_ch=[1]), dict(), dict(), dict(), dict()]
    assert load_balance_filters(bench1, 8) == ([0, 1, 0, 1, 1, 0], 2)
    # massive_config: 26 independent channels -> round robin
    massive = [dict(out_ch=[i]) for i in range(26)]
    ranks, used = load_balance_filters(massive, 8)
    assert used == 8 and ranks == [i % 8 for i in range(26)]
    # a crossbar splits by OUTPUT: the filters that mix into one output stay together, every
    # process reads all input spectra (the all-gather variant of SURVEY 8e)
    xbar = [dict(out_ch=[o]) for o in range(4) for _ in range(4)]
    assert load_balance_filters(xbar, 8) == ([o for o in range(4) for _ in range(4)], 4)
    # two islands: {0 -> 2} share nothing with {1, 3 mixing into output 5}
    isl = [dict(out_ch=[0]), dict(out_ch=[5]), dict(in_f=[0], out_ch=[1]), dict(out_ch=[5, 6])]
    assert load_balance_filters(isl, 2) == ([0, 1, 0, 1], 2)


def test_bench_self_launch_relays_rank_failure_without_a_gpu():
    """`python bench.py --gpus 2` is its own launcher (bench.py:self_launch): the parent starts
    the ranks before it touches any GPU and hands back their return code.  Without a GPU (this
    container) every rank exits with "no GPU visible": the parent must come back non-zero,
    promptly, with no result line -- never hang, never fall back to a CPU path."""
    import subprocess
    import time
    try:
        import torch
        if torch.cuda.device_count() > 0:
            pytest.skip("a GPU is visible: covered by tests/test_gpu_sharded.py")
    except ImportError:
        pass
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--workload", "B"], capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "no GPU visible" in r.stderr
    assert time.time() - t0 < 150
