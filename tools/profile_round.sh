#!/bin/bash
# Reproducible profile set of one workload for profiles/ (run ON THE GPU BOX through gpurun):
#
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh r02 C'
#
# 1. rocprofv3 --kernel-trace --stats of the very command the driver runs (python3 bench.py),
# 2. FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass:
#    MI355X_MICROARCH.md, "rocprofv3 PMC slots") over tools/pmc_driver.py (no torch in that process),
# 3. tools/make_traffic.py folds them into gpurun_out/<tag>/traffic_config<W>.json, stamped with
#    the hash of the sources the binary was built from.
# Copy gpurun_out/<tag>/*.json|csv into profiles/ afterwards (gpurun_out/ is scratch).
set -e -o pipefail
TAG=${1:-r02}
WL=${2:-C}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
EXTRA=""
if [ "$WL" != "C" ]; then EXTRA="--workload $WL"; fi

echo "[profile_round] kernel trace + stats of: python3 bench.py $EXTRA --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktrace_$WL" -- \
    python3 bench.py $EXTRA --no-cpu-baseline > "$OUT/bench_config${WL}_under_rocprofv3.json"
cat "$OUT/bench_config${WL}_under_rocprofv3.json"
STATS=$(find "$OUT/ktrace_$WL" -name '*kernel_stats.csv' | head -1)
# our kernels only (the torch RNG kernels that synthesise the inputs are not the product)
python3 - "$STATS" "$OUT/config${WL}_bench_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
keep = [rows[0]] + [r for r in rows[1:] if "bfhip::" in r[0]]
csv.writer(open(sys.argv[2], "w"), quoting=csv.QUOTE_ALL).writerows(keep)
PY
python3 profiles/summarize.py "$OUT/ktrace_$WL" 100 > "$OUT/config${WL}_bench_ktrace_summary.json"

echo "[profile_round] PMC pass 1: FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_$WL" -- \
    python3 tools/pmc_driver.py "$WL" 3 > "$OUT/pmc_fetch_$WL.log"
python3 profiles/summarize.py "$OUT/fetch_$WL" 3 > "$OUT/config${WL}_fetch_summary.json"
echo "[profile_round] PMC pass 2: WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write_$WL" -- \
    python3 tools/pmc_driver.py "$WL" 3 > "$OUT/pmc_write_$WL.log"
python3 profiles/summarize.py "$OUT/write_$WL" 3 > "$OUT/config${WL}_write_summary.json"

python3 tools/make_traffic.py "$WL" "$OUT" > "$OUT/traffic_config$WL.json"
cat "$OUT/traffic_config$WL.json"
# the heavy raw traces stay out of the merge-back
rm -rf "$OUT/ktrace_$WL" "$OUT/fetch_$WL" "$OUT/write_$WL"
echo "[profile_round] done"
