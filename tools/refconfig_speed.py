#!/usr/bin/env python3
"""Block times of the reference's own benchmark configurations (bench2/3/5, massive) on the
device: `benchmark: true` prints a per-block time in the reference (bfrun.c:2035-2078); this is
the same figure for the device backend, host buffers in and out (bfhip_engine_block)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import brutefir_amd as hip  # noqa: E402
import test_gpu_refconfigs as t  # noqa: E402


def run(name, cfg, script=None, n=200):
    e = cfg.build(hip)
    blk = t._noise_inputs(1, cfg, 1)[0]
    for b in range(20):
        if script:
            script(b, e)
        e.block(blk)
    t0 = time.perf_counter()
    for b in range(n):
        if script:
            script(b, e)
        e.block(blk)
    el = time.perf_counter() - t0
    rt = cfg.L / 44100.0
    print("%-16s %8.3f ms/block  (block = %.1f ms of audio at 44.1 kHz: %.0fx real time)" % (name, el / n * 1e3, rt * 1e3, rt / (el / n)), flush=True)


def main():
    c = t.Config(8192, 8); c.device(0, "S24_4LE", 26, range(26)); c.device(1, "S24_4LE", 26, range(26)); t._one_to_one(c, 26)
    run("bench2_config", c)
    c = t.Config(65536, 1); c.device(0, "S24_4LE", 26, range(26)); c.device(1, "S24_4LE", 26, range(26)); t._one_to_one(c, 26)
    run("bench3_config", c, n=50)
    c = t.Config(8192, 8); c.device(0, "S24_4LE", 26, range(26)); c.device(1, "S24_4LE", 26, range(26)); t._one_to_one(c, 26, crossfade=True)

    def script(b, eng):
        for f in range(26):
            eng.set_coeff(f, 0 if b % 2 == 0 else -1)
    run("bench5_config", c, script=script)
    c = t.Config(8192, 16); c.device(0, "S24_LE", 26, range(26)); c.device(1, "S24_LE", 26, range(26)); c.dither = list(range(26)); t._one_to_one(c, 26)
    run("massive_config", c)


if __name__ == "__main__":
    main()
