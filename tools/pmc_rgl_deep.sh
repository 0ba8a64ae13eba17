#!/bin/bash
# Where a wave of the RGL kernels spends its time: average in-flight vector / scalar / LDS instructions (their level counters over the
# instruction counts = latency), waves resident, the addresser's and the L1's stall reasons, instruction-cache misses.
#   usage (GPU box): bash tools/pmc_rgl_deep.sh <outdir> [isotropic|anisotropic] [lds|memory]
set -o pipefail
OUT=$(realpath -m "$1"); SHAPE=${2:-anisotropic}; SEARCH=${3:-memory}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
n=0
for pass in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
            "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TA_BUSY_sum" \
            "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/p$n" -- python3 "$REPO/tools/rgl_pmc_driver.py" $SHAPE $SEARCH > "$OUT/p$n.log" 2>&1 || { echo "pass $n failed"; tail -3 "$OUT/p$n.log"; }
done
echo ok
