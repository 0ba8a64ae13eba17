#!/bin/bash
# A/B of library builds on ONE box (devices differ by a few percent, so arms must share a box):
#   usage: bash tools/ab_libs.sh <outdir> <config> <label=ENV...>...   each arm: label then "VAR=value" assignments
set -o pipefail
OUT=$(realpath -m "$1"); CFG=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
for rep in 1 2 3; do
  for arm in "$@"; do
    label=${arm%%:*}; envs=${arm#*:}
    line=$(env $envs timeout -k 10 300 python3 "$REPO/bench.py" --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --parity-sample 256 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['value'], d['parity']['max_rel_err_vs_oracle'])") || { echo "$label failed"; exit 1; }
    echo "$CFG rep$rep $label $line" | tee -a "$OUT/ab_$CFG.txt"
  done
done
