#!/bin/bash
# does an nt brick fetch still allocate in L2 (so that coherent workloads keep their reuse)?  a 4 MB table, 99 % of the
# lookups "cold" (policy under test): if nt lines were not allocated every lookup would go to the fabric
set -o pipefail
OUT=$(realpath -m "$1"); REPO=${GRAFT_REPO_ROOT:-$(pwd)}; BIN="$REPO/tools/microbench/gather_hot"
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for mb in 2 4 16; do for aux in 0 2 18; do
  timeout -k 10 120 "$BIN" 1 1 $aux $mb >> "$OUT/timed.txt" 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d "$OUT/pmc_${mb}_$aux" -- "$BIN" 1 1 $aux $mb > "$OUT/pmc_${mb}_$aux.log" 2>&1 || exit 1
done; done
cat "$OUT/timed.txt"
