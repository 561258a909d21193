#!/bin/bash
# A/B of prebuilt library variants (_variants/lib_*.so) on one box: each is copied over the in-tree library in turn
keep=/tmp/lib_keep.so; cp lidarslam_amd/liblidarslam_amd.so $keep
for round in 1 2; do
for v in _variants/lib_*.so; do
  cp $v lidarslam_amd/liblidarslam_amd.so
  timeout -k 10 200 python scripts/match_trace_insitu.py 1 2>/dev/null | head -1 | cut -c1-90 | sed "s|^|$v |"
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   fps', round(d['value'],1), 'search us', round(d['kernels']['match_search']['us_per_launch'],1))"
done; done
cp $keep lidarslam_amd/liblidarslam_amd.so
