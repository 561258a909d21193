#!/bin/bash
# A/B on one box: the localization's start as one launch (default) against reset / undistort / boxes as separate launches
for round in 1 2 3; do
  for v in 1 0; do
    for m in 128 64; do
      timeout -k 10 200 python bench.py --model $m --no-cpu-baseline --no-extra-legs --param LocalizationStartFused=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('fused_start=$v model=$m fps', round(d['value'],1), {k:round(s[k],3) for k in ('total','undistort','submap','maps_wait')})"
    done
  done
done
