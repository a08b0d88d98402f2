#!/bin/bash
# A/B of one environment switch on ONE box: alternating bench runs (LM leg only), value + ms per step of each
# usage: scripts/ab_env.sh OUT VAR VALUE_A VALUE_B [rounds]
out=$1; var=$2; a=$3; b=$4; n=${5:-3}
: > "$out"
for i in $(seq 1 "$n"); do
  for v in "$a" "$b"; do
    env "$var=$v" timeout 300 python bench.py --no-cpu-baseline --no-frontend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', d['config'].get('lambda_search'))" >> "$out"
  done
done
cat "$out"
