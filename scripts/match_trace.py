"""Timeline of the hardware blocks of one fused match launch (VLS-128, ego-motion parameters)."""
import os, sys
os.environ["LSA_ROUTE_STATS"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lidarslam_amd as L

ctx = L.Context(0)
frames = [L.synth_frame(128, 1000, f)[0] for f in range(3)]
ctx.upload_frame(frames[0]); ctx.extract_keypoints()
ctx.upload_frame(frames[1]); c = ctx.extract_keypoints()
T = np.eye(4); T[0, 3] = 0.45
mp = L.MatchParams.ego_motion(saturation_distance=5.0)
for k, cell in ((L.EDGE, 0.5), (L.PLANE, 0.25)):
    ctx.set_target_from_set(k, L.SET_RAW_PREVIOUS, cell=cell)
for rep in range(4):
    ctx.match_types(3, L.SET_RAW_CURRENT, mp, T, slot=L.TARGET_PREVIOUS, histograms=False)
ctx.sync()
nbe, nbp = (int(c[0]) * 8 + 255) // 256, (int(c[1]) * 8 + 255) // 256
se, sp = (nbe + 7) // 8, (nbp + 7) // 8
grid = 8 * (se + sp)
tr = ctx.match_trace(grid)
ok = tr[:, 0] > 0
t0 = tr[ok, 0].min()
start, mid, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0  # us
j = np.arange(grid) // 8
is_edge = j < se
print("grid", grid, "edge blocks", nbe, "plane blocks", nbp, "span us", end[ok].max())
for name, m in (("edge", ok & is_edge), ("plane", ok & ~is_edge)):
    print(name, "blocks", m.sum(), "start p50/max", np.median(start[m]), start[m].max(), "search p50/p90/max", np.median((mid - start)[m]), np.percentile((mid - start)[m], 90), (mid - start)[m].max(),
          "end max", end[m].max())
# concurrency over time
for t in range(0, int(end[ok].max()) + 10, 10):
    print("t=%3d us running blocks %4d (edges %3d)" % (t, int(((start <= t) & (end > t) & ok).sum()), int(((start <= t) & (end > t) & ok & is_edge).sum())))
cu = (tr[:, 3] & 0xffffffff)
print("distinct (xcc, cu-ish) ids", len(set(zip((tr[ok, 3] >> 32).tolist(), ((cu[ok] >> 8) & 0xff).tolist()))))
late = np.argsort(-end * ok)[:10]
for b in late:
    print("late block", b, "edge" if is_edge[b] else "plane", "start %.1f search %.1f model %.1f end %.1f" % (start[b], mid[b] - start[b], end[b] - mid[b], end[b]), "xcc", int(tr[b, 3] >> 32))

