#!/bin/bash
# Does a physically contiguous table arena change address translation?  UTCL1 counters and read latency of the 100-table launch
# with the arena contiguous (default) and not (MRL_ARENA_CONTIGUOUS=0).   usage (GPU box): bash tools/pmc_contiguous.sh <outdir>
set -o pipefail
OUT=$(realpath -m "$1"); REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for c in 1 0; do
  export MRL_ARENA_CONTIGUOUS=$c
  for pass in "tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
              "lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum"; do
    set -- $pass; name=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/contig${c}_$name" -- \
      python3 "$REPO/bench.py" --config resident100 --steps 3 --warmup 1 --no-cpu-baseline --parity-sample 0 > "$OUT/contig${c}_$name.log" 2>&1 || { echo "pass $c $name failed"; tail -3 "$OUT/contig${c}_$name.log"; exit 1; }
  done
  mkdir -p "$OUT/c$c"; mv "$OUT"/contig${c}_* "$OUT/c$c/"
  python3 "$REPO/tools/pmc_summary.py" "$OUT/c$c" k_table_dma > "$OUT/summary_contiguous_$c.json"
done
echo ok
