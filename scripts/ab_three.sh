#!/bin/bash
# tree library against prebuilt variants (_variants/lib_*.so): frames/s and ms per ICP iteration, three alternations on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2 3; do for v in lidarslam_amd/liblidarslam_amd.so _variants/lib_*.so; do
  LSA_LIB=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --no-profile "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('$v fps', round(d['value'],1), 'ms/icp', round(d['ms_per_icp_iter'],4), 'lm', round(s['ego_lm']+s['loc_lm'],3), 'total', round(s['total'],3))"
done; done
