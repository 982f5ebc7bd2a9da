#!/usr/bin/env python3
"""Per-period time of the non-uniform partitioned convolver on BASELINE.json configs[4]
(2-in/2-out, 1 048 576-tap room correction, float64, low-latency first block).

The I/O delay of the convolver is one period of L0 frames; what has to hold for real-time use
is that EVERY period, including the ones in which the long segments are due, is processed in
less than its duration.  Prints one JSON line: median / p99 / max milliseconds per
bfhip_nupc_block call (host buffers in and out) against the period at 48 kHz, next to the
uniform engine's block time and I/O delay for the same filters."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import brutefir_amd as bf
    L0 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    seg_len, k = [], L0
    while k < 8192:
        seg_len.append(k)
        k *= 2
    seg_blk = [2] * len(seg_len)
    covered = 2 * sum(seg_len)
    seg_len.append(8192)
    seg_blk.append(-(-(1048576 - covered) // 8192))
    nu = bf.Nupc(seg_len, seg_blk, 8, 2, 2)
    nu.set_interleaved(0, "FLOAT64_LE")
    nu.set_interleaved(1, "FLOAT64_LE")
    rng = np.random.default_rng(5)
    for o in range(2):
        for i in range(2):
            h = rng.standard_normal(1048576) * np.exp(-np.arange(1048576) / 2e5) / 2000.0
            nu.add_filter(i, o, h)
    nu.finalize()
    x = rng.standard_normal((8, L0, 2)) * 0.1
    import gc
    gc.disable()                       # the timed loop allocates one small array per period
    ts = []
    for s in range(steps + 256):
        t0 = time.perf_counter()
        st, _ = nu.block(x[s & 7])
        ts.append(time.perf_counter() - t0)
        assert st == 0
    ts = np.array(ts[256:]) * 1e3
    # the periods in which every segment has a block to launch (the schedule's worst case), as
    # opposed to the rare host-side hiccups that land anywhere
    ratio = seg_len[-1] // L0
    full = ts[[i for i in range(len(ts)) if (i + 256 + 1) % ratio == 0]]
    print(json.dumps({
        "workload": "configs[4]: 2-in/2-out, %d taps, float64, partitions %s x %s" % (nu.taps, seg_len, seg_blk),
        "io_delay_frames": L0, "period_ms_at_48k": L0 / 48.0,
        "step_ms": {"median": round(float(np.median(ts)), 4), "p99": round(float(np.percentile(ts, 99)), 4),
                    "p99.9": round(float(np.percentile(ts, 99.9)), 4), "max": round(float(ts.max()), 4)},
        "all_segments_due_ms": {"periods": int(len(full)), "median": round(float(np.median(full)), 4),
                                "max": round(float(full.max()), 4)},
        "slowest_periods": [[int(i) + 256, round(float(ts[i]), 3)] for i in np.argsort(ts)[::-1][:4]],
        "realtime_margin_p99.9": round(float(L0 / 48.0 / np.percentile(ts, 99.9)), 2),
        "realtime_margin_max": round(float(L0 / 48.0 / ts.max()), 2),
        "uniform_engine_io_delay_frames": 8192, "steps": steps}), flush=True)


if __name__ == "__main__":
    main()
