#!/bin/bash
# What bounds the RGL kernels: SQ issue / wait / LDS counters and the texture addresser / L1 (TA, TCP) counters of the four entry
# points, per file shape (isotropic 8 x 32 x 32, anisotropic 16 x 8 x 32 x 32) and search mode (tables in LDS — for the anisotropic file the
# marginal rows only, and its fused call runs as eval_pdf + sample — / in memory) —
# counters only with --kernel-trace, one group per run (MI355X guide).
#   usage (GPU box): bash tools/pmc_rgl.sh <outdir>      then: python3 tools/pmc_rgl_summary.py <outdir> > profiles/r04_rgl_pmc.json
set -o pipefail
OUT=$(realpath -m "$1"); REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
run() { local cfg=$1 shape=$2 search=$3 name=$4; shift 4
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$cfg/$name" -- python3 "$REPO/tools/rgl_pmc_driver.py" $shape $search > "$OUT/$cfg.$name.log" 2>&1 || { echo "pass $cfg $name failed"; tail -5 "$OUT/$cfg.$name.log"; return 1; }
  echo "pass $cfg $name ok"; }
for cfg in isotropic:lds isotropic:memory anisotropic:lds anisotropic:memory; do
  shape=${cfg%%:*}; search=${cfg##*:}; c=${shape}_${search}
  run $c $shape $search sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD &&
  run $c $shape $search sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT &&
  run $c $shape $search ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum || exit 1
done
# the same process under --kernel-trace --stats: the per-kernel durations the rates in profiles/r04_rgl_rates.json must agree with
for cfg in isotropic:lds anisotropic:lds; do
  shape=${cfg%%:*}; search=${cfg##*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_${shape}_${search}" -- python3 "$REPO/tools/rgl_pmc_driver.py" $shape $search > "$OUT/stats_${shape}_${search}.log" 2>&1 || { echo "stats $cfg failed"; exit 1; }
  find "$OUT/stats_${shape}_${search}" -name "*kernel_stats.csv" -exec cp {} "$OUT/rgl_${shape}_${search}_kernel_stats.csv" \;
done
echo ok
