#!/usr/bin/env python3
"""create / use / destroy engines of every flavour a few hundred times and watch the device's free
memory: a leak shows as a steady drift"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import brutefir_amd as bf  # noqa: E402


def free_mib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20


def one(k):
    rng = np.random.default_rng(k)
    L = int(rng.choice([64, 256, 8192, 16384]))
    rs = int(rng.choice([4, 8]))
    e = bf.Engine(L, 3, rs, 2, 3)
    e.set_interleaved(0, "S24_4LE")
    e.set_interleaved(1, "S16_LE")
    if L <= 8192:
        e.set_powersave(1.0)
    h = (rng.standard_normal(L * 3) / 100).astype(np.float32 if rs == 4 else np.float64)
    c = e.add_coeff(h)
    e.add_filter(in_ch=[0], out_ch=[0], coeff=c, crossfade=True)
    e.add_filter(in_ch=[0, 1], out_ch=[1], coeff=c)
    e.add_filter(in_f=[0], out_ch=[2], coeff=-1)
    e.enable_dither([0], 44100, 0)
    e.finalize()
    x = (rng.standard_normal((L, 2)) * 1e5).astype(np.int32)
    e.block(x)
    if L <= 8192:
        e.rt_begin(k & 1)
        e.rt_block(x)
        e.set_coeff(0, -1)
        e.rt_block(x)
        e.rt_end()
    e.close()
    if k % 7 == 0:
        nu = bf.Nupc([64, 128, 256], [2, 2, 4], rs, 1, 1)
        nu.set_interleaved(0, "FLOAT_LE")
        nu.set_interleaved(1, "FLOAT_LE" if rs == 4 else "FLOAT64_LE")
        nu.add_filter(0, 0, (rng.standard_normal(1000) / 30).astype(np.float32 if rs == 4 else np.float64))
        nu.finalize()
        for _ in range(5):
            nu.block(rng.standard_normal((64, 1)).astype(np.float32))
        nu.close()


def two(k):
    """the round-2 machinery: coefficient slabs + reservation, the stream-ordered copy, the wave FFT,
    deferred / ping-pong block schedules with outputs still owed when the engine is destroyed"""
    rng = np.random.default_rng(1000 + k)
    L = int(rng.choice([4096, 8192]))
    rs = int(rng.choice([4, 8]))
    sched = k % 3
    os.environ["BFHIP_COEFF_STREAM"] = "2"
    os.environ["BFHIP_MAC_TARGET_WGS"] = "32" if L == 8192 else "16"
    if sched == 0:
        os.environ["BFHIP_OVERLAP"] = "0"
    elif sched == 1:
        os.environ["BFHIP_OVERLAP"] = "1"
    try:
        e = bf.Engine(L, 2, rs, 8, 8)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "S24_4LE")
        e.reserve_coeffs(16 * 2 * 2 * L * rs)          # room for a quarter of the sets: the rest spills
        dt = np.float32 if rs == 4 else np.float64
        for o in range(8):
            for i in range(8):
                h = (rng.standard_normal(2 * L) / 400).astype(dt)
                e.add_filter(in_ch=[i], out_ch=[o], coeff=e.add_coeff(h))
        e.finalize()
    finally:
        for v in ("BFHIP_COEFF_STREAM", "BFHIP_MAC_TARGET_WGS", "BFHIP_OVERLAP"):
            os.environ.pop(v, None)
    x = torch.from_numpy((rng.standard_normal((L, 8)) * 1e5).astype(np.int32)).cuda()
    y = torch.zeros(L, 8, dtype=torch.int32, device="cuda")
    for _ in range(4):
        e.block_dev(x, y)
    e.update_coeff_block(3, 1, (rng.standard_normal(L) / 400).astype(dt))
    e.set_coeff(5, 9)
    e.block_dev(x, y)
    if k & 1:
        assert e.sync() == 0
    e.close()                                           # with outputs still owed on the even turns
    del x, y


def long_run(n_blocks=3000, shared_inputs=True):
    """ONE engine that lives long: a control change on every block (coefficient toggles with
    cross-fades, scale and block-delay changes -- the first of each gives the filter a private ring
    --, sample-delay changes on N:1 channels, a coefficient partition refreshed in place), first
    through the block call, then in real-time mode, where every plan change re-captures the
    graphs.  Device AND host memory must stay where they are after the first few hundred blocks."""
    import resource
    rng = np.random.default_rng(77)
    L, N = 256, 4
    e = bf.Engine(L, N, 4, 3, 3)
    if shared_inputs:
        e.map_channels(0, [0, 1, 1])                      # inputs 1 and 2 share a physical channel:
        e.set_interleaved_phys(0, "S24_4LE", 2)           # per-block job tables, no graph replay
        for v in (1, 2):
            e.set_maxdelay(0, v, 600)
    else:
        e.set_interleaved(0, "S24_4LE")                   # replayable: plan changes re-capture the graphs
    e.set_interleaved(1, "S24_4LE")
    cs = [e.add_coeff((rng.standard_normal(L * N) / 200).astype(np.float32)) for _ in range(3)]
    e.add_filter(in_ch=[0], out_ch=[0], coeff=cs[0], crossfade=True)
    e.add_filter(in_ch=[1], out_ch=[1], coeff=cs[1])
    e.add_filter(in_ch=[0, 2], out_ch=[2], coeff=cs[2])
    e.add_filter(in_f=[1], out_ch=[0], coeff=-1)
    e.finalize()
    x = (rng.standard_normal((L, 2 if shared_inputs else 3)) * 1e5).astype(np.int32)
    marks = []

    def control(k):
        if not shared_inputs and k % 3:
            return                                        # leave blocks for the graphs to replay
        e.set_coeff(0, cs[(k // 3) % 2])
        if k % 3 == 0:
            e.set_scale(1, 0, 0, 1.0 + (k % 5) * 0.1)
        if k % 7 == 0:
            e.set_delayblocks(1, k % N)
        if shared_inputs and k % 5 == 0:
            e.set_delay(0, 1 + k % 2, (37 * k) % 600)
        if k % 11 == 0:
            e.update_coeff_block(cs[2], k % N, (rng.standard_normal(L) / 200).astype(np.float32))

    for k in range(n_blocks):
        if k == n_blocks // 2:
            e.rt_begin(0)
        control(k)
        st, _ = e.block(x) if k < n_blocks // 2 else e.rt_block(x)
        assert st == 0, (k, st)
        if k % 500 == 499:
            marks.append((free_mib(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0))
    stats = e.rt_stats()
    e.rt_end()
    e.close()
    print("long run (%s): %d blocks, %d graph captures, %d replayed; (device free MiB, host max RSS MiB) every 500 blocks: %s"
          % ("N:1 inputs" if shared_inputs else "replayable", n_blocks, stats["captures"], stats["graph"],
             ["%.1f/%.0f" % m for m in marks]))
    assert shared_inputs or stats["captures"] > 100
    assert abs(marks[0][0] - marks[-1][0]) < 8, "device memory moves with the number of blocks"
    assert marks[-1][1] - marks[1][1] < 64, "host memory grows with the number of blocks"


def main():
    long_run(shared_inputs=True)
    long_run(shared_inputs=False)
    for k in range(20):
        one(k)
        two(k)
    base = free_mib()
    marks = []
    for k in range(20, 320):
        one(k)
        if k % 4 == 0:
            two(k)
        if k % 100 == 19:
            marks.append(free_mib())
    print("free MiB after warm-up %.1f, then %s -> drift %.1f MiB over 300 engines" % (base, ["%.1f" % m for m in marks], base - marks[-1]))
    assert base - marks[-1] < 64, "device memory is leaking"


if __name__ == "__main__":
    main()
