#!/bin/bash
# A/B of two prebuilt libraries on one box: scripts/ab_libs.sh _variants/lib_a.so _variants/lib_b.so [models...]
A=$1; B=$2; shift 2; MODELS=${@:-128 64}
keep=/tmp/lib_keep.so; cp lidarslam_amd/liblidarslam_amd.so $keep
for round in 1 2 3; do
  for v in $A $B; do
    cp $v lidarslam_amd/liblidarslam_amd.so
    for m in $MODELS; do
      timeout -k 10 200 python bench.py --model $m --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$v model=$m fps', round(d['value'],1), 'ms/icp', round(d['ms_per_icp_iter'],4), {n:round(v['us_per_launch'],1) for n,v in k.items() if n.startswith('match') or n=='lm_solve'})"
    done
  done
done
cp $keep lidarslam_amd/liblidarslam_amd.so
