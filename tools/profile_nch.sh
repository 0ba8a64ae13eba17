#!/bin/bash
# rocprofv3 --kernel-trace --stats of the n-channel rate tool (kernel names + average durations per width)
set -o pipefail
OUT=$(realpath -m "$1")
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/nch_stats" -- python3 "$REPO/tools/nch_rates.py" > "$OUT/nch_rates.json" 2> "$OUT/nch.err" || { echo "nch stats failed"; tail -5 "$OUT/nch.err"; exit 1; }
find "$OUT/nch_stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/nch_kernel_stats.csv" \;
rm -rf "$OUT/nch_stats"
echo ok
