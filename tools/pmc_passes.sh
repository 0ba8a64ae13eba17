#!/bin/bash
# Collects rocprofv3 PMC counters for the bench kernel in separate passes (counters only with
# --kernel-trace; one counter group per run, as the MI355X guide prescribes).
#   usage (on the GPU box, via gpurun):  bash tools/pmc_passes.sh <outdir> [bench args...]
set -o pipefail
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU &&
run sq2 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY &&
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum &&
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum &&
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum &&
run ta1 TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run grbm GRBM_GUI_ACTIVE
