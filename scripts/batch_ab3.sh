#!/bin/bash
# Diagnostic: 8 sequences side by side, maps on the host / on the device; then one sequence (bench), alternating map streams
for maps in 0 1 0 1; do
timeout -k 10 300 python - <<PY
import sys
sys.path.insert(0, ".")
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed
L.bind_host_to_device(0)
rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range(8)], 40, lookahead=True, EgoMotion=3, MapsOnDevice=$maps)
fps = rep.run(8)
rep.close()
print("maps on device $maps S=8 fps", round(fps, 1), flush=True)
PY
done
bash scripts/ab_variants.sh "head||" "own|LSA_MAP_STREAM=own|" "shared|LSA_MAP_STREAM=shared|" | head -6
