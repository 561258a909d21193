#!/bin/bash
# prebuilt library variants (_variants/lib_*.so, LSA_LIB) on one box: frame rate and the per-launch time of the named scopes
# usage: scripts/ab_variants_kernels.sh "scope1 scope2" [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=$1; shift
for round in 1 2; do for v in lidarslam_amd/liblidarslam_amd.so _variants/lib_*.so; do
  LSA_LIB=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$v fps', round(d['value'],1), {n:round(k[n]['us_per_launch'],1) for n in '$S'.split() if n in k})"
done; done
