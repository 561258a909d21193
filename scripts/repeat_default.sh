#!/bin/bash
# the driver's command line (profiling scopes on, extra legs off here) against --no-profile, interleaved, N times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq ${1:-4}); do for extra in "" "--no-profile"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('[$extra] fps', round(d['value'],1), 'ms/icp', round(d['ms_per_icp_iter'],4), {k:round(v,3) for k,v in s.items()}, (d.get('roofline') or {}).get('kernel'))"
done; done
