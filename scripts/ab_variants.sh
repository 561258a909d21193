#!/bin/bash
# A/B on ONE box (boxes of the pool differ by more than most changes): alternates variants given as "NAME|ENV|ARGS", 3 rounds
run() { env $2 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('$1', 'fps', round(d['value'],1), 'ego_lm', round(s['ego_lm'],3), 'loc_lm', round(s['loc_lm'],3), 'submap', round(s['submap'],3), 'total', round(s['total'],3))"; }
for round in 1 2 3; do
  for v in "$@"; do IFS='|' read -r name envs args <<< "$v"; run "$name" "${envs:-X=1}" "$args"; done
done
