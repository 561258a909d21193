"""One point of the batch sweep: S sequences side by side (host maps unless MAPS=1), prints aggregate frames/s.  Environment knobs
(LSA_LM_CACHE, LSA_LM_RECORDS, LSA_LM_BLOCKS) are read by the library at context creation."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L.bind_host_to_device(0)
rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range(S)], 40, lookahead=True, EgoMotion=3, MapsOnDevice=int(os.environ.get("MAPS", "0")))
fps = rep.run(8)
rep.close()
print(json.dumps({"S": S, "fps": round(fps, 1), "env": {k: v for k, v in os.environ.items() if k.startswith("LSA_") or k == "MAPS"}}), flush=True)
