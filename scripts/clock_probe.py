"""Diagnostic: frame rate of a resident replay with and without a featherweight load beside it (clock management)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import lidarslam_amd as L

def replay(busy_blocks, frames=48, warm=8):
    slam = L.Slam(0, EgoMotion=3)
    stamps = []
    for f in range(frames):
        pts, stamp = L.synth_frame(128, 1000, f)
        slam.store_frame(f, pts)
        stamps.append(stamp)
    ctx = slam.context()
    t0 = None
    for f in range(frames):
        if f == warm:
            ctx.sync()
            if busy_blocks:
                ctx._check(ctx.L.lsa_selftest_keep_busy(ctx.h, 150, busy_blocks), "keep_busy")
                time.sleep(0.02)
            t0 = time.perf_counter()
        if f + 1 < frames:
            slam.hint_next_stored_frame(f + 1)
        slam.add_stored_frame(f, stamps[f], f)
    ctx.sync()
    dt = time.perf_counter() - t0
    time.sleep(0.3)
    slam.close()
    return (frames - warm) / dt

for blocks in (0, 8, 0, 64, 0, 256):
    print("busy blocks", blocks, "fps", round(replay(blocks), 1), flush=True)
