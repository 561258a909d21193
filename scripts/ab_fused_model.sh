#!/bin/bash
# A/B on one box: the model fits inside the search kernel (default) against the two-launch form (LSA_FUSED_MODEL=0)
for round in 1 2 3; do
  for v in 1 0; do
    for m in 128 64 16; do
      LSA_FUSED_MODEL=$v timeout -k 10 200 python bench.py --model $m --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('fused_model=$v model=$m fps', round(d['value'],1), 'ms/icp', round(d['ms_per_icp_iter'],4), {n:round(v['us_per_launch'],1) for n,v in k.items() if n.startswith('match') or n=='lm_solve'})"
    done
  done
done
