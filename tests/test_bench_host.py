"""CPU: the host-side pieces of bench.py that run without a GPU -- the CPU-baseline worker (the
oracle timed on a bounded sample of the workload), the source hash that ties profiles/ to a
binary, the core count -- on small shapes, for every raw format a workload uses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_every_workload_format_has_a_cpu_baseline_input():
    formats = {wl[5] for wl in bench.WORKLOADS.values()}
    assert formats <= {"S24_4LE", "FLOAT64_LE"}, formats        # extend _cpu_worker with the workload


@pytest.mark.parametrize("wl", [(3, 2, 64, 3, 4, "S24_4LE"), (2, 2, 64, 5, 8, "FLOAT64_LE")])
def test_cpu_worker_feeds_the_oracle_buffers_of_the_right_size(wl):
    """float64 workloads (E, F) once got int32 frames: the oracle read past the buffer"""
    rate, n_blocks, setup_s = bench._cpu_worker((wl, 0, 0.2))
    assert rate > 0 and n_blocks >= 1 and setup_s >= 0
    with pytest.raises(SystemExit):
        bench._cpu_worker(((2, 2, 64, 2, 4, "S16_LE"), 0, 0.1))


def test_cpu_worker_child_process_prints_one_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", "1", "--workload", "B",
                        "--cpu-seconds", "0.2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rate, n_blocks, _ = json.loads(r.stdout.strip().splitlines()[-1])
    assert rate > 0 and n_blocks >= 1


def test_source_hash_names_the_sources_of_the_binary():
    h = bench.source_hash()
    assert len(h) == 16 and int(h, 16) >= 0
    for w in ("C", "F"):                 # the committed PMC traffic belongs to these very sources
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_config%s.json" % w)))
        if tj["source_hash"] != h:       # bench.py then reports traffic_stale: true; not a failure of the code
            pytest.skip("profiles/traffic_config%s.json was measured on other sources (%s, now %s): "
                        "re-run tools/profile_round.sh" % (w, tj["source_hash"], h))


def test_usable_cores_is_bounded_by_the_affinity_mask():
    cores, note = bench.usable_cores()
    assert 1 <= cores <= len(os.sched_getaffinity(0)) and isinstance(note, str)
