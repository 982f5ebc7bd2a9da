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
