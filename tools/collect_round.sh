#!/bin/bash
# Everything the round's committed evidence is made of, on one box and one build: per-config bench lines + kernel stats + fabric
# counters, the VALU counters of the headline kernel, the RGL counters / rates / L2 counters, the two parity soaks, the gather
# microbenchmark.   usage (GPU box): bash tools/collect_round.sh <outdir>      then tools/collect_profiles.py etc.
set -o pipefail
OUT=$(realpath -m "$1"); REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd "$REPO"
bash tools/profile_configs.sh "$OUT/configs" merl64m ggx64m mixed16_256m resident100 > "$OUT/configs.log" 2>&1 || { echo "configs failed"; tail -5 "$OUT/configs.log"; exit 1; }
echo "configs ok"
bash tools/pmc_valu.sh "$OUT/valu" > "$OUT/valu.log" 2>&1 || { echo "valu failed"; exit 1; }
echo "valu ok"
bash tools/pmc_rgl.sh "$OUT/rgl" > "$OUT/rgl.log" 2>&1 || { echo "rgl pmc failed"; exit 1; }
echo "rgl pmc ok"
bash tools/pmc_rgl_l2.sh "$OUT/rgl_l2" > "$OUT/rgl_l2.log" 2>&1 || echo "rgl l2 failed"
cd "$REPO"
timeout -k 10 300 python3 tools/rgl_rates.py > "$OUT/rgl_rates.json" 2> "$OUT/rgl_rates.err" || { echo "rates failed"; exit 1; }
echo "rates ok"
timeout -k 10 600 python3 tools/fuzz_parity_rgl.py 240 > "$OUT/fuzz_parity_rgl.json" 2> "$OUT/fuzz_parity_rgl.err" || { echo "rgl soak failed"; tail -3 "$OUT/fuzz_parity_rgl.err"; exit 1; }
echo "rgl soak ok"
timeout -k 10 600 python3 tools/fuzz_parity.py 96 > "$OUT/fuzz_parity.json" 2> "$OUT/fuzz_parity.err" || { echo "table soak failed"; tail -3 "$OUT/fuzz_parity.err"; exit 1; }
echo "table soak ok"
if [ -x tools/microbench/gather_quad ]; then
  for mb in 1 13 36; do for m in scatter lane64 quadraw quad lane32 pair32; do timeout -k 10 60 tools/microbench/gather_quad $mb $m 4 || exit 1; done; done > "$OUT/gather_quad.jsonl" 2>&1
  echo "microbench ok"
fi
echo done
