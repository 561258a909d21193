#!/bin/bash
# S = 8 sequences on one GPU (device maps) under a few runtime settings: hardware queues, workgroups of the one-launch solve
cat > /tmp/b8.py <<'PY'
import os, sys, time, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed
L.bind_host_to_device(0)
S = int(sys.argv[1]); maps = int(sys.argv[2])
rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range(S)], 40, lookahead=True, EgoMotion=3, MapsOnDevice=maps)
fps = rep.run(8)
fb = [int(s.get_param("DeviceSolveFallbacks")) for s in rep.slams]; gt = [int(s.get_param("IcpGateTimeouts")) for s in rep.slams]
rep.close()
print("S", S, "maps", "device" if maps else "host", os.environ.get("GPU_MAX_HW_QUEUES", "-"), os.environ.get("LSA_LM_BLOCKS", "-"), os.environ.get("LSA_ICP_AHEAD", "-"), "fps", round(fps, 1), "fallbacks", sum(fb), "gate timeouts", sum(gt))
PY
for q in 4 8; do for lmb in 64 32 16; do for ah in 1 0; do
  GPU_MAX_HW_QUEUES=$q LSA_LM_BLOCKS=$lmb LSA_ICP_AHEAD=$ah timeout -k 10 200 python /tmp/b8.py 8 1 2>&1 | tail -1
done; done; done
