"""Diagnostic: sizes of the matching problem of the headline workload (queries, targets per type) and route statistics."""
import sys
import numpy as np
sys.path.insert(0, ".")
import lidarslam_amd as L
slam = L.Slam(0, EgoMotion=3)
for f in range(24):
    pts, stamp = L.synth_frame(128, 1000, f)
    slam.add_frame(pts, stamp, f)
ctx = slam.context()
print("points", pts.size)
for k in range(3):
    print("type", k, "keypoints", slam.keypoints(k).size, "submap", slam.target_submap(k).size, "map", slam.map(k).size,
          "prev target", ctx.L.lsa_target_size(ctx.h, 0, k))
st = slam.stats()
print("stats", [round(x, 5) for x in st])
