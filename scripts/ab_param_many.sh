#!/bin/bash
# one pipeline parameter at two values, N alternations of the headline leg on one box: scripts/ab_param_many.sh Name A B N
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=$1; A=$2; B=$3; N=${4:-8}
for i in $(seq $N); do for v in $A $B; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --param $P=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('$P=$v fps', round(d['value'],1), 'total', round(s['total'],3), 'ego_lm', round(s['ego_lm'],3), 'submap', round(s['submap'],3), 'maps_wait', round(s['maps_wait'],3))"
done; done
