#!/bin/bash
# sweep of the first kNN kernel's shape and the grid cells (results are identical, only the speed changes)
run() { python bench.py --steps 30 --warmup 8 --cpu-frames 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; s=d['stage_ms_per_frame']
print(' '.join(a for a in sys.argv[1:] if a!='--param'), '| fps %.1f'%d['value'], ' '.join('%s=%.0f'%(n.replace('knn_',''),k[n]['us_per_launch']) for n in ('knn_fine_edge','knn_coarse_edge','knn_fine_plane','knn_coarse_plane','model_edge','model_plane') if n in k), 'ego %.2f loc %.2f'%(s['ego_icp']+s['ego_lm'], s['loc_icp']+s['loc_lm']), flush=True)" "$@"; }
run
run --param KnnRoundsEdges=3
run
run --param KnnLanesEdges=8
