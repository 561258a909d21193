#!/bin/bash
# sweep of the kNN grid cell sizes (results are identical, only the speed changes)
run() { python bench.py --steps 15 --warmup 5 --cpu-frames 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print(' '.join(sys.argv[1:]), '| fps %.1f'%d['value'], ' '.join('%s=%.0f'%(n,k[n]['us_per_launch']) for n in ('knn_edge','knn_plane','target_grid_build') if n in k), 'ego_icp %.2f loc_icp %.2f'%(d['stage_ms_per_frame']['ego_icp'], d['stage_ms_per_frame']['loc_icp']))" "$@"; }
for ee in 0.5 1.0 1.5; do run --param KnnCellSizeEgoMotionEdges=$ee; done
for me in 1.5 2.5 4.0; do run --param KnnCellScaleMapsEdges=$me; done
for mp in 0.5 1.0 1.5; do run --param KnnCellScaleMaps=$mp; done
for ep in 0.2 0.35; do run --param KnnCellSizeEgoMotion=$ep; done
