#!/bin/bash
# A/B over the three sensors on one box: "NAME|ARGS" variants
for m in 128 64 16; do for v in "$@"; do IFS='|' read -r name args <<< "$v"
timeout -k 10 300 python bench.py --model $m --no-cpu-baseline --no-extra-legs $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('model $m $name', 'fps', round(d['value'],1), 'ego_lm', round(s['ego_lm'],3), 'loc_lm', round(s['loc_lm'],3), 'submap', round(s['submap'],3), 'maps_wait', round(s['maps_wait'],3))"
done; done
