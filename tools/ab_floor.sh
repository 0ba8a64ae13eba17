#!/bin/bash
# A/B of library builds on ONE box, on the bench's random inputs AND on coherent inputs (bench.py --coherent 65536: table
# traffic L2-served, the launch shows its VALU / LDS floor).  Arms share the box because devices differ by a few percent.
#   usage: bash tools/ab_floor.sh <outfile> <label=path/to/libmerl_hip.so>...     (label "head" = the in-tree build)
set -o pipefail
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$(dirname "$OUT")"
for rep in 1 2 3; do
  for arm in "$@"; do
    label=${arm%%=*}; lib=${arm#*=}
    for mode in random coherent; do
      if [ $mode = coherent ]; then A="--coherent 65536"; else A=""; fi
      if [ "$lib" = head ]; then E=""; else E="MRL_LIB_PATH=$REPO/$lib"; fi
      line=$(env $E timeout -k 10 300 python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-scalar-calls --parity-sample 4096 $A $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['value'], d['parity']['max_rel_err_vs_oracle'], d['parity']['values_beyond_tolerance'])") || { echo "$label $mode failed"; exit 1; }
      echo "rep$rep $label $mode $line" | tee -a "$OUT"
    done
  done
done
