#!/bin/bash
# A/B of one boolean pipeline parameter on one box: scripts/ab_param.sh Name [models...]
P=$1; shift; MODELS=${@:-128 64 16}
for round in 1 2 3; do
  for v in 1 0; do
    for m in $MODELS; do
      timeout -k 10 200 python bench.py --model $m --no-cpu-baseline --no-extra-legs --param $P=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('$P=$v model=$m fps', round(d['value'],1), 'ms/icp', round(d['ms_per_icp_iter'],4), {k:round(s[k],3) for k in ('total','ego_lm','submap','maps_wait')})"
    done
  done
done
