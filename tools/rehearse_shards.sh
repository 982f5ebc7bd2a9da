#!/bin/bash
# One rank's share of an N-GPU run of config C measured alone on ONE GPU, for both ways of splitting
# the crossbar (bench.py --shard input | output), N = 2 4 8, all on the same box:
#   gpurun --timeout 900 -- 'bash tools/rehearse_shards.sh r03'
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
FAILED=0
for n in 2 4 8; do
  for mode in input output; do
    name=configC_rank0of${n}_${mode}_sharded_rehearsal
    BFHIP_BENCH_REHEARSE_RANKS=$n python bench.py --shard $mode --no-cpu-baseline 2> "$OUT/$name.err" | tail -1 > "$OUT/bench_$name.json"
    rc=${PIPESTATUS[0]}
    if [ "$rc" != 0 ] || [ ! -s "$OUT/bench_$name.json" ]; then mv -f "$OUT/bench_$name.json" "$OUT/bench_$name.json.FAILED" 2>/dev/null; FAILED=$((FAILED + 1)); fi
    python3 - "$OUT/bench_$name.json" "$n" "$mode" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read())
    print("N=%s %-6s block %.4f ms  MAC %.4f ms frac %.3f  io %.4f ms  verify %s" % (sys.argv[2], sys.argv[3], j["ms_per_step"],
          j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["roofline"]["fft_in_ms"], j["verify"]["ok"]))
except Exception as ex:
    print("N=%s %s FAILED %s" % (sys.argv[2], sys.argv[3], ex))
PY
  done
done
[ "$FAILED" = 0 ] || { echo "rehearse_shards: $FAILED run(s) FAILED"; exit 1; }
