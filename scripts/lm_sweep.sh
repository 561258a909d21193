#!/bin/bash
# Diagnostic: the solve kernel's time for several splits of the residual blocks over workgroups
for cfg in "1024 64" "512 64" "512 128" "384 128" "256 128"; do
  set -- $cfg
  LSA_LM_RECORDS=$1 LSA_LM_BLOCKS=$2 timeout -k 10 200 python scripts/match_trace_insitu.py 1 2>/dev/null | grep "LM solves" | sed "s/^/records $1 blocks $2: /"
  LSA_LM_RECORDS=$1 LSA_LM_BLOCKS=$2 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   fps', round(d['value'],1), 'lm_solve us', round(d['kernels']['lm_solve']['us_per_launch'],1))"
done
