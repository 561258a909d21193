#!/bin/bash
# Diagnostic: S sequences on one GPU: maps on the device / on the host, host threads bound to the GPU's NUMA node or not
nproc; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; echo "GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-unset}"
for cfg in "1 1 4" "1 1 8" "0 1 4" "0 1 8" "1 0 8" "0 0 8"; do
  set -- $cfg
  timeout -k 10 300 python - <<PY
import os, sys, json, time
sys.path.insert(0, ".")
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed
if $2: L.bind_host_to_device(0)
rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range($3)], 40, lookahead=True, EgoMotion=3, MapsOnDevice=$1)
t0 = time.process_time()
fps = rep.run(8)
cpu = time.process_time() - t0
rep.close()
print("maps on device $1 numa-bound $2 S=$3 fps", round(fps, 1), "cpu cores busy", round(cpu / (32 * $3 / fps), 1), flush=True)
PY
done
