#!/bin/bash
# Soaks of the reference-loop parity tests (run ON THE GPU BOX through gpurun, one call per line; each
# fits one call's time limit).  The default test run takes 24 seeds of every family; a soak takes hundreds
# from a seed offset, so that successive soaks cover new cases:
#
#   gpurun --timeout 1150 -- 'bash tools/soak_refloop.sh combined 1000 700'
#   gpurun --timeout 1150 -- 'bash tools/soak_refloop.sh channels 1000 600'
#   gpurun --timeout 1150 -- 'bash tools/soak_refloop.sh networks 0 500'
#   gpurun --timeout 1150 -- 'bash tools/soak_refloop.sh engine 0 1500'      # engine vs oracle fuzz + shards
#
# Output: gpurun_out/soak/<family>_<seed0>.txt; exit code non-zero on any failure or GPU memory fault.
FAMILY=${1:-combined}; SEED0=${2:-0}; N=${3:-300}
mkdir -p gpurun_out/soak
LOG=gpurun_out/soak/${FAMILY}_${SEED0}.txt
case "$FAMILY" in
  combined) K="random_networks_over"; FILES=tests/test_gpu_refloop.py ;;
  channels) K="shared_channels"; FILES=tests/test_gpu_refloop.py ;;
  networks) K="over_the_product or patched_filter_process"; FILES=tests/test_gpu_refloop.py ;;      # (the seed offset moves the patched multi-process cases only)
  engine)   K=""; FILES="tests/test_gpu_fuzz.py tests/test_gpu_shards.py" ;;
  *) echo "unknown family $FAMILY"; exit 2 ;;
esac
export BFHIP_REFLOOP_SEED0=$SEED0 BFHIP_REFLOOP_SEEDS=$N BFHIP_FUZZ_SEEDS=$N BFHIP_SHARD_SEEDS=$((N / 2))
if [ -n "$K" ]; then timeout -k 10 1100 python -m pytest $FILES -q -k "$K" > "$LOG" 2>&1; else timeout -k 10 1100 python -m pytest $FILES -q > "$LOG" 2>&1; fi
RC=$?
tail -5 "$LOG"
if grep -q "Memory access fault" "$LOG"; then echo "GPU MEMORY FAULT in $LOG"; exit 1; fi
exit $RC
