#!/bin/bash
# like ab.sh, profiling scopes off (pure wall clock)
run() { LSA_LIB=$1 python bench.py --steps 30 --warmup 8 --cpu-frames 0 --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_frame']
print(sys.argv[1], 'fps %.1f'%d['value'], ' '.join('%s=%.3f'%(n,v) for n,v in s.items() if n not in ('maps_wait','total')), flush=True)" $2; }
for i in 1 2 3 4; do run lidarslam_amd/_ab/liblidarslam_amd.so base; run "" new; done
