"""Aggregate frames/s of S independent sequences side by side on one GPU (frame store + look-ahead), for a few settings."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lidarslam_amd as L
from lidarslam_amd.replay import ConcurrentReplay, sequence_seed

L.bind_host_to_device(0)
frames, warm = 40, 8
for params in ({"MapsOnDevice": 1}, {"MapsOnDevice": 0}):
    for S in (1, 2, 4, 8):
        rep = ConcurrentReplay(0, 128, [sequence_seed(s) for s in range(S)], frames, lookahead=True, EgoMotion=3, **params)
        t0 = time.process_time()
        fps = rep.run(warm)
        cpu = time.process_time() - t0
        rep.close()
        print(json.dumps({"params": params, "S": S, "fps": round(fps, 1), "per_seq": round(fps / S, 1), "cpu_s_per_wall_s": round(cpu / ((frames - warm) * S / fps), 2)}), flush=True)
