#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or counter collection) per kernel.

    python profiles/summarize.py <dir with *_kernel_trace.csv / *_counter_collection.csv> [last_k]

Durations are averaged over the LAST `last_k` dispatches of every kernel (default 50) so that
the engine's warm-up blocks -- which skip partitions that do not exist yet, exactly like the
reference's `procblocks` guard (bfrun.c:1745) -- do not dilute the steady-state figure.
Counter values are per dispatch; FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them.
On gfx950 FETCH_SIZE counts exactly half of a wide coalesced read stream
(MI355X_MICROARCH.md, HBM section): `fetch_bytes_corrected` = 2 * 1024 * FETCH_SIZE.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    last_k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in per.items():
            if "bfhip::" not in k:
                continue
            name = k.split("(")[0].replace("void ", "")
            tail = v[-last_k:]
            out.setdefault(name, {}).update({
                "calls": len(v), "avg_ns_all": sum(v) / len(v),
                "steady_calls": len(tail), "avg_ns_steady": sum(tail) / len(tail),
                "min_ns": min(v), "max_ns": max(v)})
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in per.items():
            if "bfhip::" not in k:
                continue
            name = k.split("(")[0].replace("void ", "")
            tail = v[-min(last_k, 3):]
            ent = out.setdefault(name, {})
            ent[c + "_per_dispatch_steady"] = sum(tail) / len(tail)
            if c == "FETCH_SIZE":
                ent["fetch_bytes_corrected"] = 2 * 1024 * sum(tail) / len(tail)
            if c == "WRITE_SIZE":
                ent["write_bytes"] = 1024 * sum(tail) / len(tail)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
