#!/bin/bash
# A/B on one box: RefineUndistortion between two localization iterations inside the next search kernel (default) or as a launch of its own
for round in 1 2 3; do
  for v in 1 0; do
    for m in 128 64; do
      timeout -k 10 200 python bench.py --model $m --no-cpu-baseline --no-extra-legs --param UndistortInSearch=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('undistort_in_search=$v model=$m fps', round(d['value'],1), 'ms/icp', round(d['ms_per_icp_iter'],4), {k:round(s[k],3) for k in ('total','loc_icp','loc_lm')})"
    done
  done
done
