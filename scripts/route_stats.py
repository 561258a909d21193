"""Which routes do the searches of the fused match kernel take?  (VLS-128 frames, ego-motion and localization-like targets)"""
import os, sys
os.environ["LSA_ROUTE_STATS"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lidarslam_amd as L

ctx = L.Context(0)
ctx.profile(True)
frames = [L.synth_frame(128, 1000, f)[0] for f in range(3)]
ctx.upload_frame(frames[0]); ctx.extract_keypoints()
ctx.upload_frame(frames[1]); c = ctx.extract_keypoints()
T = np.eye(4); T[0, 3] = 0.45
names = ["tail", "tail done", "second scans", "first block > shell 2", "walked", "far by counts", "first block = shell 0", "longest walk"]
for name, mp, cells in (("ego", L.MatchParams.ego_motion(saturation_distance=5.0), (0.5, 0.25)), ("loc", L.MatchParams.localization(saturation_distance=2.0), (0.75, 0.6))):
    for k in (L.EDGE, L.PLANE):
        ctx.set_target_from_set(k, L.SET_RAW_PREVIOUS, cell=cells[k])
    for rep in range(5):
        ctx.match_types(3, L.SET_RAW_CURRENT, mp, T, slot=L.TARGET_PREVIOUS, histograms=False)
    ctx.sync()
    for k in (L.EDGE, L.PLANE):
        rs = ctx.route_stats(k)
        print(name, "type", k, "queries", int(c[k]), dict(zip(names, rs.tolist())))
for st in ctx.profile_stats():
    print(st["name"], st["launches"], round(1e3 * st["total_ms"] / max(st["launches"], 1), 1), "us")
