#!/bin/bash
# Runs tools/microbench/gather_partial for every request width, first timed, then under rocprofv3 PMC passes
# (counters only with --kernel-trace, one group per run).   usage: bash tools/partial_line_probe.sh <outdir>
set -o pipefail
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
BIN="$REPO/tools/microbench/gather_partial"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1 || true
grep -o "TCC_EA0_[A-Z0-9_]*" "$OUT/counters.txt" | sort -u > "$OUT/tcc_ea0_counters.txt" || true
for lpl in 8 6 60 4 2 0; do
  timeout -k 10 120 "$BIN" 187 $lpl >> "$OUT/timed.txt" 2>&1 || { echo "timed run lpl=$lpl failed"; exit 1; }
done
cat "$OUT/timed.txt"
for lpl in 8 6 60 4 2 0; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum --output-format csv -d "$OUT/req_$lpl" -- "$BIN" 187 $lpl > "$OUT/req_$lpl.log" 2>&1 || { echo "pmc req lpl=$lpl failed"; tail -3 "$OUT/req_$lpl.log"; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$lpl" -- "$BIN" 187 $lpl > "$OUT/fetch_$lpl.log" 2>&1 || { echo "pmc fetch lpl=$lpl failed"; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write_$lpl" -- "$BIN" 187 $lpl > "$OUT/write_$lpl.log" 2>&1 || { echo "pmc write lpl=$lpl failed"; exit 1; }
  echo "pmc lpl=$lpl ok"
done
