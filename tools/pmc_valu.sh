#!/bin/bash
# VALU evidence for the dominant kernel: SQ instruction / busy counters of the bench launch, on the
# bench's random inputs and on coherent inputs (bench.py --coherent 65536: table traffic L2-served, the launch sits on its
# non-fabric floor).  Counters only with --kernel-trace, one group per run (MI355X guide).
#   usage (GPU box): bash tools/pmc_valu.sh <outdir> [extra bench args]
set -o pipefail
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench-args, counters...
  local name=$1; local bargs=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 $bargs $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
for mode in random coherent; do
  if [ $mode = coherent ]; then A="--coherent 65536"; else A=""; fi
  run ${mode}_sq1 "$A" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS &&
  run ${mode}_sq2 "$A" SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS &&
  run ${mode}_sq3 "$A" SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 || exit 1
done
python3 "$REPO/tools/pmc_summary.py" "$OUT" k_table_dma > "$OUT/summary.json" || true
