"""Diagnostic: which stage of AddFrame waits when an unrelated kernel occupies a side stream."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import lidarslam_amd as L

names = ["total", "extract", "ego_icp", "ego_lm", "loc_icp", "loc_lm", "undistort", "submap", "maps"]
params = dict(kv.split("=") for kv in sys.argv[1:])
slam = L.Slam(0, EgoMotion=3, **{k: float(v) for k, v in params.items()})
frames = 20
stamps = []
for f in range(frames):
    pts, stamp = L.synth_frame(128, 1000, f)
    slam.store_frame(f, pts)
    stamps.append(stamp)
ctx = slam.context()
for f in range(frames):
    if f == 10:
        ctx.sync()
        ctx._check(ctx.L.lsa_selftest_keep_busy(ctx.h, 60, 8), "keep_busy")
    if f + 1 < frames:
        slam.hint_next_stored_frame(f + 1)
    t0 = time.perf_counter()
    slam.add_stored_frame(f, stamps[f], f)
    dt = time.perf_counter() - t0
    st = slam.stats()
    print(f, round(1e3 * dt, 2), {n: round(1e3 * st[i], 2) for i, n in enumerate(names) if st[i] > 2e-4}, flush=True)
ctx.sync()
slam.close()
