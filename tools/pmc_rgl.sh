#!/bin/bash
# What bounds the RGL kernels: SQ issue / wait counters and the texture addresser / L1 (TA, TCP) busy counters of
# tools/rgl_rates.py's launches (counters only with --kernel-trace, one group per run).
#   usage (GPU box): bash tools/pmc_rgl.sh <outdir>
set -o pipefail
OUT=$(realpath -m "$1"); REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/tools/rgl_rates.py" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD &&
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F64 &&
run ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
python3 "$REPO/tools/pmc_summary.py" "$OUT" k_rgl > "$OUT/summary.json" || true
echo ok
