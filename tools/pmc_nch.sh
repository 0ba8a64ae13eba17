#!/bin/bash
# fabric requests of the n-channel kernels, one width per run (TCC_EA0 requests + L2 hits, counters only with --kernel-trace)
set -o pipefail
OUT=$(realpath -m "$1"); REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for C in 1 4 8 16 32; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/nch_$C" -- python3 "$REPO/tools/nch_rates.py" $C > "$OUT/nch_$C.log" 2>&1 || { echo "pmc nch $C failed"; tail -3 "$OUT/nch_$C.log"; exit 1; }
done
echo ok
