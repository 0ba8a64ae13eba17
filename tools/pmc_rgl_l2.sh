#!/bin/bash
# The L2 side of the RGL kernels: hit / miss / fabric read requests and fetched bytes per launch, per file shape (tools/rgl_pmc_driver.py).
#   usage (GPU box): bash tools/pmc_rgl_l2.sh <outdir>
set -o pipefail
OUT=$(realpath -m "$1"); REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for shape in isotropic anisotropic; do
  for pass in "hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" "fetch FETCH_SIZE" "lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
    set -- $pass; name=$1; shift
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$shape/$name" -- python3 "$REPO/tools/rgl_pmc_driver.py" $shape memory > "$OUT/$shape.$name.log" 2>&1 || { echo "pass $shape $name failed"; tail -5 "$OUT/$shape.$name.log"; }
  done
done
echo ok
