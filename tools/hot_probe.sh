#!/bin/bash
set -o pipefail
OUT=$(realpath -m "$1"); REPO=${GRAFT_REPO_ROOT:-$(pwd)}; BIN="$REPO/tools/microbench/gather_hot"
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for hot in "3 15" "3 50" "24 60"; do for aux in 0 2 16 18 1 3; do
  timeout -k 10 120 "$BIN" $hot $aux >> "$OUT/timed.txt" 2>&1 || { echo "timed $hot $aux failed"; exit 1; }
done; done
cat "$OUT/timed.txt"
for hot in "3 15" "3 50"; do for aux in 0 2 16 18; do
  set -- $hot
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d "$OUT/pmc_$1_$2_$aux" -- "$BIN" $1 $2 $aux > "$OUT/pmc_$1_$2_$aux.log" 2>&1 || { echo "pmc failed"; exit 1; }
done; done
echo pmc ok
