#!/usr/bin/env python3
"""Fold the two rocprofv3 PMC passes and the kernel-trace stats of one workload (written by
tools/profile_round.sh) into traffic_config<W>.json, the file bench.py reads `roofline.traffic`
from.  The file names the sources it was measured on (`source_hash`, bench.py:source_hash) so
that bench.py can flag it when it no longer describes the binary being timed.

    python3 tools/make_traffic.py C gpurun_out/r02 > gpurun_out/r02/traffic_configC.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_hash only; nothing here touches a GPU)


def main():
    wl, d = sys.argv[1], sys.argv[2]
    fetch = json.load(open(os.path.join(d, "config%s_fetch_summary.json" % wl)))
    write = json.load(open(os.path.join(d, "config%s_write_summary.json" % wl)))
    kt = json.load(open(os.path.join(d, "config%s_bench_ktrace_summary.json" % wl)))
    live = json.loads([ln for ln in open(os.path.join(d, "bench_config%s_under_rocprofv3.json" % wl))
                       if ln.startswith("{")][-1])
    # the MAC of the plan: the crossbar kernel, or mac_diag_kernel for one-to-one plans (workload D)
    macs = [k for k in kt if ("mac_xbar_kernel" in k or "mac_diag_kernel" in k) and k in fetch and k in write]
    name = max(macs, key=lambda k: kt[k]["avg_ns_all"] * kt[k]["calls"])
    f, w, k = fetch[name], write[name], kt[name]
    fetch_b = f["fetch_bytes_corrected"]
    write_b = w["write_bytes"]
    alg = live["roofline"]["algorithmic_bytes_per_launch"]
    commit = os.environ.get("BFHIP_COMMIT", "")
    out = {
        "kernel": name, "workload": wl,
        "source_hash": bench.source_hash(),
        "commit": commit or None,
        "fetch_size_kib_per_launch": f["FETCH_SIZE_per_dispatch_steady"],
        "fetch_bytes_corrected_x2": fetch_b,
        "write_bytes": write_b,
        "traffic_bytes_per_launch": fetch_b + write_b,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (fetch_b + write_b) / alg,
        "rocprof_avg_launch_ns": {"all_launches": k["avg_ns_all"], "calls": k["calls"],
                                  "last_100": k["avg_ns_steady"]},
        "live_avg_launch_ms_same_run": live["roofline"]["avg_launch_ms"],
        "live_frac_same_run": live["roofline"]["frac"],
        "rocprof_frac": alg / (k["avg_ns_all"] * 1e-9) / 1e9 / bench.HBM_PEAK_GBS,
        "pmc_frac_of_peak_at_rocprof_time": (fetch_b + write_b) / (k["avg_ns_all"] * 1e-9) / 1e9 / bench.HBM_PEAK_GBS,
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_driver.py "
                  "(steady-state launches), KiB*1024, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts "
                  "half of a wide coalesced read stream); durations from rocprofv3 --kernel-trace --stats of "
                  "`python3 bench.py` (every launch prewarmed = steady state); tools/profile_round.sh",
    }
    print(json.dumps(out, indent=1))
    # the bench line of the profiled run was printed before these counters existed: it could only
    # see the traffic file of an earlier binary (and said so: traffic_stale).  Same call, same
    # binary, same box -- give it the figures measured beside it.
    path = os.path.join(d, "bench_config%s_under_rocprofv3.json" % wl)
    if live.get("source_hash") == out["source_hash"]:
        live["roofline"].update({"traffic": out["traffic_bytes_per_launch"], "traffic_stale": False,
                                 "traffic_source_hash": out["source_hash"],
                                 "traffic_note": "PMC passes of the same tools/profile_round.sh call"})
        open(path, "w").write(json.dumps(live) + "\n")


if __name__ == "__main__":
    main()
