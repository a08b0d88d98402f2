#!/bin/bash
# A/B of environment knobs on one box: scripts/ab_env.sh OUT CONFIGS ROUNDS "ENV=VAL ..." "ENV=VAL ..." ...   ("-" = the default)
out=$1; cfgs=$2; rounds=$3; shift 3
for cfg in $cfgs; do
  for i in $(seq $rounds); do
    for v in "$@"; do
      ev=$v; [ "$v" = "-" ] && ev=""
      env $ev python bench.py --config $cfg --no-cpu-baseline --no-frontend 2> /dev/null | V="$v" C=$cfg python -c '
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
q=d["config"]["lambda_search"]
print("config %s %-28s %.1f it/s  %.4f ms/step  chol %.2f us  solves %d/%d  err %.12g" % (os.environ["C"], os.environ["V"]+":", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], q["solves_used"], q["solves_queued"], d["config"]["error_after"]))' >> $out
    done
  done
done
cat $out
