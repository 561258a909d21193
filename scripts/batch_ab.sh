#!/bin/bash
# Diagnostic: 8 sequences on one GPU (host maps, the batch default) for several settings
for cfg in "X=1" "LSA_PREP_EARLY=1" "X=1" "LSA_PREP_EARLY=1"; do
  env $cfg timeout -k 10 300 python - <<PY
import os, sys, json
sys.path.insert(0, ".")
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed
L.bind_host_to_device(0)
rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range(8)], 40, lookahead=True, EgoMotion=3)
fps = rep.run(8)
rep.close()
print("$cfg", "S=8 fps", round(fps, 1), flush=True)
PY
done
