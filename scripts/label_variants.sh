#!/bin/bash
# k_label time (HIP events, causal replay) for the tree library and the _variants
for v in tree _variants/lib_*.so; do
  [ "$v" = tree ] && unset LSA_LIB || export LSA_LIB=$v
  timeout -k 10 200 python bench.py --causal --steps 30 --warmup 6 --no-cpu-baseline --no-extra-legs --profile-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$v', 'fps', round(d['value'],1), {n:round(k[n]['us_per_launch'],1) for n in ('label_nms','curvature','ring_bucket','compact','invalidate') if n in k})"
done
