#!/bin/bash
# A/B of an environment switch on one box: scripts/ab_env.sh VAR A B [bench args]   (three alternations)
V=$1; A=$2; B=$3; shift 3
for round in 1 2 3; do
  for x in $A $B; do
    env $V=$x timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('$V=$x fps', round(d['value'],1), 'ego_lm+loc_lm', round(s['ego_lm']+s['loc_lm'],3), 'icp', round(s['ego_icp']+s['loc_icp'],3), 'total', round(s['total'],3), d['config'].get('icp_gate_timeouts'))"
  done
done
