#!/bin/bash
# Why does the 100-resident-table launch (BASELINE configs[4]) run at 87 % of the 19 GB gather microbenchmark, and why does it
# move +-5 % from one process to the next?    Five plain runs for the spread, then counter passes on
# address translation (UTCL1), the share of fabric reads that reach DRAM, and read latency — for the single-table launch,
# the 16-table launch and the 100-table launch side by side.
#   usage (GPU box): bash tools/pmc_resident.sh <outdir>
set -o pipefail
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  timeout -k 10 300 python3 "$REPO/bench.py" --config resident100 --steps 20 --warmup 3 --no-cpu-baseline --parity-sample 0 $BENCH_ARGS 2>/dev/null \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('resident100 run $i kernel_ms', d['roofline']['kernel_ms'], 'Meval/s', d['value'])" | tee -a "$OUT/spread.txt" || exit 1
done
run() {  # name, config, counters...
  local name=$1; local cfg=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/${cfg}_$name" -- \
    python3 "$REPO/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 $BENCH_ARGS > "$OUT/${cfg}_$name.log" 2>&1 || { echo "pass $cfg $name failed"; tail -5 "$OUT/${cfg}_$name.log"; return 1; }
  echo "pass $cfg $name ok"
}
for cfg in merl64m mixed16_256m resident100; do
  run tlb $cfg TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum &&
  run tlbstall $cfg TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum &&
  run dram $cfg TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum &&
  run lat $cfg TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum || exit 1
done
python3 "$REPO/tools/pmc_summary.py" "$OUT" k_table_dma > "$OUT/summary.json" || true
