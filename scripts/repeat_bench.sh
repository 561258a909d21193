#!/bin/bash
# the headline leg N times on one box (run-to-run spread), optionally under an environment setting: scripts/repeat_bench.sh N [VAR=VAL ...] [-- bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=$1; shift
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for i in $(seq $N); do
  env "${ENVS[@]}" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --no-profile "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
acc=sum(v for k,v in s.items() if k not in ('total','maps_async','maps_wait'))
print('${ENVS[*]} fps', round(d['value'],1), 'ms/icp', round(d['ms_per_icp_iter'],4), 'total', round(s['total'],3), 'outside the loops', round(s['total']-s['ego_lm']-s['loc_lm']-s['ego_icp']-s['loc_icp'],3), 'extract', round(s['extract'],3), 'unaccounted', round(s['total']-acc,3), 'adopted', d['config'].get('extractions_taken_over'), 'fallbacks', d['config'].get('device_solve_fallbacks'), {k:round(v,3) for k,v in s.items() if k in ('undistort','submap','maps','maps_wait','maps_async','extract')})"
done
