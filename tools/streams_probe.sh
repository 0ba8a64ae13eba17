#!/bin/bash
# tools/microbench/gather_streams for every (load, store) cache-policy pair, timed and under one PMC pass.
set -o pipefail
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
BIN="$REPO/tools/microbench/gather_streams"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lf in 0 1 2; do for sf in 0 1 2 3; do
  timeout -k 10 120 "$BIN" 187 $lf $sf >> "$OUT/timed.txt" 2>&1 || { echo "timed $lf $sf failed"; exit 1; }
done; done
cat "$OUT/timed.txt"
for pair in "1 1" "1 2" "1 3" "2 2" "0 0" "2 3"; do
  set -- $pair
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_$1_$2" -- "$BIN" 187 $1 $2 > "$OUT/pmc_$1_$2.log" 2>&1 || { echo "pmc $pair failed"; exit 1; }
done
echo "pmc ok"
