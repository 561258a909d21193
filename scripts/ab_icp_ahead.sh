#!/bin/bash
# ICP loops in line (0), behind gates (1), enqueued whole behind links (2): one box, three alternations, then 8 sequences side by side
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in 0 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --no-profile --param ICPAhead=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']
print('ICPAhead=$v fps', round(d['value'],1), 'ms/icp', round(d['ms_per_icp_iter'],4), {k:round(s[k],3) for k in ('total','ego_icp','ego_lm','loc_icp','loc_lm')})"
  done
done
cat > /tmp/b8.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed
L.bind_host_to_device(0)
S = int(sys.argv[1]); maps = int(sys.argv[2]); ahead = int(sys.argv[3])
rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range(S)], 40, lookahead=True, EgoMotion=3, MapsOnDevice=maps, ICPAhead=ahead)
fps = rep.run(8)
fb = [int(s.get_param("DeviceSolveFallbacks")) for s in rep.slams]
rep.close()
print("S", S, "maps", "device" if maps else "host", "ICPAhead", ahead, "fps", round(fps, 1), "fallbacks", sum(fb))
PY
for round in 1 2; do for maps in 1 0; do for ah in 0 2; do
  timeout -k 10 200 python /tmp/b8.py 8 $maps $ah 2>&1 | tail -1
done; done; done
for s in 2 4; do timeout -k 10 200 python /tmp/b8.py $s 1 2 2>&1 | tail -1; done
