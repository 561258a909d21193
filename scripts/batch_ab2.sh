#!/bin/bash
# Diagnostic: same box, several trees (worktrees of earlier commits beside the current one), 8 sequences, host maps
for d in "$@"; do
  ( cd $d && timeout -k 10 300 python - <<PY
import os, sys, json, time
sys.path.insert(0, ".")
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed
L.bind_host_to_device(0)
rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range(8)], 40, lookahead=True, EgoMotion=3, MapsOnDevice=0)
fps = rep.run(8)
rep.close()
print("tree $d host maps S=8 fps", round(fps, 1), flush=True)
PY
  )
done
