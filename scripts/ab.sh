#!/bin/bash
# A/B on one box: lidarslam_amd/_ab/liblidarslam_amd.so (baseline, built by scripts/ab_baseline.sh from HEAD) against
# the working-tree build, interleaved runs.
run() { LSA_LIB=$1 python bench.py --steps 30 --warmup 8 --cpu-frames 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_frame']; k=d['kernels']
print(sys.argv[1], 'fps %.1f'%d['value'], 'spec %.2f'%d.get('submap_speculation_hits_per_frame',-1), ' '.join('%s=%.3f'%(n,v) for n,v in s.items() if n not in ('maps_wait','total')), '|', ' '.join('%s=%.0f'%(n.replace('knn_',''),k[n]['us_per_launch']) for n in ('knn_fine_edge','knn_coarse_edge','knn_fine_plane','knn_coarse_plane','accumulate_jac','label_nms') if n in k), flush=True)" $2; }
for i in 1 2 3; do run lidarslam_amd/_ab/liblidarslam_amd.so base; run "" new; done
