"""CPU-restatement baseline at several OpenMP thread counts."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lidarslam_amd as L
from oracle import oracle as O
model = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
frames = [L.synth_frame(model, 1000, f) for f in range(nframes + 2)]
out = {"model": model, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"), "affinity": len(os.sched_getaffinity(0))}
for th in (1, 4, 16):
    s = O.Slam(EgoMotion=3, NbThreads=th)
    ts, stats = [], np.zeros(16)
    for f, (pts, stamp) in enumerate(frames):
        t = time.perf_counter(); s.add_frame(pts, stamp, f); dt = time.perf_counter() - t
        if f >= 2:
            ts.append(dt); stats += s.stats()
    ts = np.array(ts)
    iters = max(stats[9] + stats[10], 1)
    out[f"threads_{th}"] = {"fps": float(len(ts) / ts.sum()), "median_ms": float(1e3 * np.median(ts)), "p95_ms": float(1e3 * np.percentile(ts, 95)),
                            "ms_per_icp_iter": float(1e3 * (stats[2] + stats[3] + stats[4] + stats[5]) / iters),
                            "stage_ms": {k: float(1e3 * stats[i] / len(ts)) for i, k in enumerate(["total", "extract", "ego_icp", "ego_lm", "loc_icp", "loc_lm", "undistort", "submap", "maps"])}}
    print(th, out[f"threads_{th}"], flush=True)
print(json.dumps(out))
