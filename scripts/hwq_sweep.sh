#!/bin/bash
# Diagnostic: one sequence (bench) and 8 sequences (host maps / device maps) for several numbers of hardware queues
for q in 4 8 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hw queues $q: one sequence fps', round(d['value'],1))"
  timeout -k 10 300 python - <<PY
import os, sys
sys.path.insert(0, ".")
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed
L.bind_host_to_device(0)
for maps in (0, 1):
    rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range(8)], 40, lookahead=True, EgoMotion=3, MapsOnDevice=maps)
    fps = rep.run(8)
    rep.close()
    print("hw queues $q: 8 sequences, maps on device", maps, "fps", round(fps, 1), flush=True)
PY
done
