#!/usr/bin/env python3
"""print the interesting fields of bench.py JSON lines (files given as arguments, else stdin)"""
import json
import sys

import fileinput

for line in fileinput.input():
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    print("%.4f ms/step  %.1f Msamples/s  whole-block %s GB/s | mac %.4f ms = %s GB/s (%.1f%%)  fft_in %.4f  ifft_out %.4f" % (
        d["ms_per_step"], d["value"] / 1e6, round(d.get("hbm_gbs_algorithmic") or 0),
        r.get("avg_launch_ms", 0), round(r.get("achieved") or 0), 100 * (r.get("frac") or 0),
        r.get("fft_in_ms", 0), r.get("ifft_out_ms", 0)))
