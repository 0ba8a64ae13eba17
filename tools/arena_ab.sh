set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r03
for rep in 1 2 3 4 5; do
 for arm in 0 20480; do
  timeout -k 10 300 python3 bench.py --config resident100 --steps 20 --warmup 3 --no-cpu-baseline --parity-sample 256 --arena-mb $arm 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('resident100 arena_mb $arm rep $rep kernel_ms', d['roofline']['kernel_ms'], d['value'], d['parity']['values_beyond_tolerance'])" | tee -a gpurun_out/r03/arena_ab.txt || exit 1
 done
done
for arm in 0 4096; do
  timeout -k 10 300 python3 bench.py --config mixed16_256m --steps 10 --warmup 2 --no-cpu-baseline --parity-sample 256 --arena-mb $arm 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('mixed16 arena_mb $arm kernel_ms', d['roofline']['kernel_ms'], d['value'])" | tee -a gpurun_out/r03/arena_ab.txt || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum --output-format csv -d $REPO/gpurun_out/r03/arena_tlb -- python3 $REPO/bench.py --config resident100 --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 --arena-mb 20480 > $REPO/gpurun_out/r03/arena_tlb.log 2>&1 && python3 $REPO/tools/pmc_summary.py $REPO/gpurun_out/r03 k_table_dma | head -0; echo done
