#!/bin/bash
# quick sweep of the kNN grid cell sizes (results are identical, only the speed changes)
for ego in 0.25 0.5 1.0; do for ms in 0.5 1.0 2.0; do
  python bench.py --steps 10 --warmup 4 --cpu-frames 0 --param KnnCellSizeEgoMotion=$ego --param KnnCellScaleMaps=$ms 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('ego',$ego,'maps',$ms,'fps %.1f'%d['value'], ' '.join('%s=%.0f'%(n,k[n]['us_per_launch']) for n in ('knn_edge','knn_plane','target_grid_build','accumulate_jac') if n in k), d['stage_ms_per_frame'])"
done; done
