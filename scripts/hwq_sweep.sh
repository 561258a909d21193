#!/bin/bash
# Diagnostic: aggregate frame rate of 8 sequences on one GPU for several numbers of hardware queues of the process
for q in 2 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python - <<PY
import os, sys, json
sys.path.insert(0, ".")
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed
L.bind_host_to_device(0)
for params in ({}, {"MapsOnDevice": 0}):
    rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range(8)], 40, lookahead=True, EgoMotion=3, **params)
    fps = rep.run(8)
    rep.close()
    print("hw queues", os.environ["GPU_MAX_HW_QUEUES"], params, "S=8 fps", round(fps, 1), flush=True)
PY
done
