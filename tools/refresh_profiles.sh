#!/bin/bash
# Re-measure every bench line kept under profiles/ on the current sources (run ON THE GPU BOX):
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh r03'
# then copy gpurun_out/<tag>/bench_*.json to profiles/<tag>_bench_*.json.  The kernel statistics and
# PMC traffic of configs C and F come from tools/profile_round.sh.
# A run that fails (a verify mismatch makes bench.py exit non-zero AFTER printing its line) leaves
# bench_<name>.json.FAILED, never a plausible-looking bench_<name>.json, and the script's own exit
# code is non-zero when any run failed.
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
FAILED=0
run() {
  name=$1; shift
  "$@" 2> "$OUT/$name.err" | tail -1 > "$OUT/bench_$name.json"
  rc=${PIPESTATUS[0]}
  if [ "$rc" != 0 ] || [ ! -s "$OUT/bench_$name.json" ]; then
    mv -f "$OUT/bench_$name.json" "$OUT/bench_$name.json.FAILED" 2>/dev/null
    FAILED=$((FAILED + 1))
  fi
  echo "$name rc=$rc"
}
run configC python bench.py
run configF python bench.py --workload F
for w in B D E C2 C4 C8; do run config$w python bench.py --workload $w --no-cpu-baseline; done
run configC_hostio python bench.py --host-io --no-cpu-baseline
for n in 2 4 8; do BFHIP_BENCH_REHEARSE_RANKS=$n run configC_rank0of${n}_rehearsal python bench.py --no-cpu-baseline; done
# the reference's own process rule as the multi-GPU split: rank r owns O/N outputs, no collective
for n in 2 4 8; do BFHIP_BENCH_REHEARSE_RANKS=$n run configC_rank0of${n}_output_sharded_rehearsal python bench.py --shard output --no-cpu-baseline; done
BFHIP_DIST_BACKEND=gloo run configC_2rank_output_sharded_gloo python bench.py --gpus 2 --shard output --steps 40 --warmup 10 --no-cpu-baseline
# informative: two blocks per pass over the coefficients (bfhip_engine_block_pair_dev)
run configC_pairs_informative python bench.py --pairs --no-cpu-baseline
run configF_pairs_informative python bench.py --pairs --workload F --no-cpu-baseline
BFHIP_BENCH_REHEARSE_RANKS=8 BFHIP_BENCH_REHEARSE_RCCL=1 run configC_rank0of8_rccl_one_rank python bench.py --no-cpu-baseline
BFHIP_BENCH_REHEARSE_RANKS=8 BFHIP_BENCH_REHEARSE_RCCL=1 run configD_rank0of8_rccl_one_rank python bench.py --workload D --no-cpu-baseline
BFHIP_DIST_BACKEND=gloo run configC_2rank_gloo_rehearsal python bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline
BFHIP_DIST_BACKEND=gloo run configD_2ranks_gloo_one_gpu python bench.py --gpus 2 --workload D --steps 40 --warmup 10 --no-cpu-baseline
if [ "$FAILED" != 0 ]; then echo "refresh_profiles: $FAILED run(s) FAILED"; exit 1; fi
