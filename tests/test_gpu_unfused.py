"""GPU: the UNFUSED block loop (examples/bfunfused.c) -- a C host that keeps filter_process()'s own
per-block structure and calls the convolver.h symbols of libbfhip.so one buffer at a time, with
the per-buffer events of struct bfevents (bfmod.h:192-215) fired at the reference's call sites
(bfrun.c:1533, 1554, 1688, 1839, 1882, 1918).  This is the path INTEGRATION.md keeps for
configurations whose logic modules register such an event.

Checked against the oracle engine: without hooks sample for sample, and with a module on every one
of the six events.  The module multiplies the buffer it is handed by a power of two, which is
layout independent and commutes exactly with every later step, so each event has an equivalent
plain configuration the oracle can run:
    input_timed / input_freqd on input i  == that factor on the input scale of every filter fed by i
    pre_convolve on filter f              == the factor on f's input scale
    post_convolve on filter f             == the factor on all partitions of f but the first (the
                                             ring slot is scaled AFTER the block that filled it)
    output_freqd / output_timed on out o  == the factor on the output scale of every filter into o
"""
import os
import re
import subprocess

import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, N, I, O = 256, 3, 3, 2
EVENTS = ["input_timed", "input_freqd", "pre_convolve", "post_convolve", "output_freqd", "output_timed"]


@pytest.fixture(scope="module")
def exe(hip):
    path = os.path.join(ROOT, "examples", "bfunfused")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "bfunfused.c"), "-o", path,
                           "-L" + os.path.join(ROOT, "brutefir_amd"), "-lbfhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "brutefir_amd")])
    return path


def run_host(exe, tmp_path, rs, irs, raw, hooks=()):
    dt = np.float32 if rs == 4 else np.float64
    np.stack([irs[(o, i)] for o in range(O) for i in range(I)]).astype(dt).tofile(tmp_path / "coeffs.bin")
    raw.tofile(tmp_path / "in.raw")
    r = subprocess.run([exe, str(rs), str(L), str(N), str(I), str(O), str(tmp_path / "coeffs.bin"),
                        str(tmp_path / "in.raw"), str(tmp_path / "out.raw")] + list(hooks),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    calls = {e: int(re.search(e + r" (\d+)", r.stderr).group(1)) for e in EVENTS}
    return np.fromfile(tmp_path / "out.raw", np.int32).reshape(-1, O), calls, r.stderr


def run_oracle(rs, irs, blocks, in_scale=None, out_scale=None):
    dt = np.float32 if rs == 4 else np.float64
    oe = bo.Engine(L, N, rs, I, O)
    oe.set_interleaved(0, "S24_4LE")
    oe.set_interleaved(1, "S24_4LE")
    for o in range(O):
        for i in range(I):
            oe.add_filter(in_ch=[i], out_ch=[o], coeff=oe.add_coeff(irs[(o, i)].astype(dt)),
                          in_scale=[1.0 if in_scale is None else in_scale[o][i]],
                          out_scale=[1.0 if out_scale is None else out_scale[o][i]])
    outs = []
    for b in blocks:
        st, out = oe.block(b)
        assert st == 0
        outs.append(out.view(np.int32).reshape(L, O).copy())
    return np.concatenate(outs)


def make(rs, seed):
    dt = np.float32 if rs == 4 else np.float64
    irs = {(o, i): cases.make_ir(np.random.default_rng(seed + o * I + i), L * N, I).astype(dt)
           for o in range(O) for i in range(I)}
    blocks = cases.raw_blocks(seed + 100, N + 5, L, I, "S24_4LE", amplitude=0.05)
    return irs, blocks


@pytest.mark.parametrize("rs", [4, 8])
def test_unfused_loop_over_the_convolver_h_symbols_equals_the_oracle(hip, exe, tmp_path, rs):
    irs, blocks = make(rs, 500)
    raw = np.concatenate(blocks)
    got, calls, log = run_host(exe, tmp_path, rs, irs, raw)
    want = run_oracle(rs, irs, blocks)
    assert "%d blocks" % len(blocks) in log and all(v == 0 for v in calls.values())
    assert got.shape == want.shape
    assert np.abs(got.astype(np.int64) - want).max() <= 1           # S24 LSB: the FFTs differ in rounding
    # ... and the fused engine of the same network
    ge, _ = cases.crossbar(hip.Engine, L, N, rs, I, O, "S24_4LE", "S24_4LE", seed=500)
    fused = np.concatenate([np.frombuffer(ge.block(b)[1].tobytes(), np.int32).reshape(L, O) for b in blocks])
    assert np.abs(got.astype(np.int64) - fused).max() <= 1


@pytest.mark.parametrize("rs", [4, 8])
def test_every_per_buffer_event_reaches_its_module_and_may_rewrite_the_buffer(hip, exe, tmp_path, rs):
    irs, blocks = make(rs, 700)
    raw = np.concatenate(blocks)
    hooks = ["input_timed:0:0.5", "input_freqd:1:4", "pre_convolve:2:2", "post_convolve:4:0.25",
             "output_freqd:0:-1", "output_timed:1:0.5"]
    got, calls, _ = run_host(exe, tmp_path, rs, irs, raw, hooks)
    nb = len(blocks)
    assert calls == {"input_timed": nb * I, "input_freqd": nb * I, "pre_convolve": nb * I * O,
                     "post_convolve": nb * I * O, "output_freqd": nb * O, "output_timed": nb * O}
    # the equivalent plain configuration (filter f = o * I + i)
    in_scale = [[1.0] * I for _ in range(O)]
    out_scale = [[1.0] * I for _ in range(O)]
    for o in range(O):
        in_scale[o][0] *= 0.5                          # input_timed on input 0
        in_scale[o][1] *= 4.0                          # input_freqd on input 1
    in_scale[2 // I][2 % I] *= 2.0                     # pre_convolve on filter 2
    irs2 = dict(irs)
    tail = irs[(4 // I, 4 % I)].copy()
    tail[L:] *= 0.25                                   # post_convolve on filter 4: partitions 1.. only
    irs2[(4 // I, 4 % I)] = tail
    for i in range(I):
        out_scale[0][i] *= -1.0                        # output_freqd on output 0
        out_scale[1][i] *= 0.5                         # output_timed on output 1
    want = run_oracle(rs, irs2, blocks, in_scale, out_scale)
    plain = run_oracle(rs, irs, blocks)
    assert np.abs(want.astype(np.int64) - plain).max() > 1000          # the hooks change the result ...
    assert np.abs(got.astype(np.int64) - want).max() <= 1               # ... exactly as predicted
