#!/bin/bash
# Per BASELINE config: the bench line, rocprofv3 --kernel-trace --stats, and the three PMC passes that give the
# fabric bytes per launch (TCC_EA0 requests, FETCH_SIZE, WRITE_SIZE — one group per run, counters only with --kernel-trace).
#   usage (GPU box): bash tools/profile_configs.sh <outdir> <config> [config...]
set -o pipefail
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  echo "== $cfg"
  timeout -k 10 400 python3 "$REPO/bench.py" --config $cfg --steps 20 --warmup 3 > "$OUT/${cfg}_bench.json" 2> "$OUT/${cfg}_bench.err" || { echo "bench $cfg failed"; tail -5 "$OUT/${cfg}_bench.err"; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${cfg}_stats" -- python3 "$REPO/bench.py" --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${cfg}_stats.log" 2>&1 || { echo "stats $cfg failed"; tail -5 "$OUT/${cfg}_stats.log"; exit 1; }
  for pass in "req TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "fetch FETCH_SIZE" "write WRITE_SIZE" "hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
    set -- $pass; name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/${cfg}_pmc/$name" -- python3 "$REPO/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 > "$OUT/${cfg}_pmc_$name.log" 2>&1 || { echo "pmc $name $cfg failed"; tail -5 "$OUT/${cfg}_pmc_$name.log"; exit 1; }
  done
  python3 "$REPO/tools/pmc_summary.py" "$OUT/${cfg}_pmc" mrl > "$OUT/${cfg}_pmc_summary.json" 2>&1 || true
  find "$OUT/${cfg}_stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/${cfg}_kernel_stats.csv" \;
  echo "$cfg ok"
done
