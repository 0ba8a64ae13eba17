#!/bin/bash
# Does the chip hold its clock under this kernel?  Runs bench.py for a few thousand steps in the background and samples
# rocm-smi (sclk, power, temperature) twice a second meanwhile.   usage: bash tools/clock_probe.sh <outfile> [bench args]
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$(dirname "$OUT")"
timeout -k 10 200 python3 "$REPO/bench.py" --steps 6000 --warmup 3 --no-cpu-baseline --no-scalar-calls --parity-sample 0 "$@" > "$OUT.bench.json" 2>/dev/null &
BP=$!
: > "$OUT"
for i in $(seq 1 80); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | tr -d '\n' >> "$OUT"; echo >> "$OUT"
  sleep 0.25
done
wait $BP
cat "$OUT.bench.json" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms', d['roofline']['kernel_ms'])"
python3 - "$OUT" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    c=d.get('card0',{})
    print({k:v for k,v in c.items() if 'sclk' in k.lower() or 'power' in k.lower() or 'Temperature (Sensor junction)' in k or 'mclk' in k.lower()})
PY
