set -e
mkdir -p gpurun_out/r02c
for w in B D E; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/r02c/bench_config$w.json 2>/dev/null; echo "$w rc=$?"; done
python bench.py --host-io --no-cpu-baseline > gpurun_out/r02c/bench_configC_hostio.json 2>/dev/null; echo "hostio rc=$?"
for n in 2 4 8; do BFHIP_BENCH_REHEARSE_RANKS=$n python bench.py --no-cpu-baseline > gpurun_out/r02c/bench_configC_rank0of${n}_rehearsal.json 2>/dev/null; echo "rehearse $n rc=$?"; done
for w in C2 C4 C8; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/r02c/bench_config$w.json 2>/dev/null; echo "$w rc=$?"; done
BFHIP_BENCH_REHEARSE_RANKS=8 BFHIP_BENCH_REHEARSE_RCCL=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02c/bench_configC_rank0of8_rccl_one_rank.json; echo "rccl rc=$?"
BFHIP_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r02c/bench_configC_2rank_gloo_rehearsal.json 2>/dev/null; echo "gloo2 rc=$?"
python tools/nupc_latency.py > gpurun_out/r02c/nupc_latency.json 2> gpurun_out/r02c/nupc_latency.err; echo "nupc rc=$?"
