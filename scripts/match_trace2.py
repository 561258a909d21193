"""Per-type block timings of the fused match launch: planes alone, edges alone, both (VLS-128, ego-motion parameters)."""
import os, sys
os.environ["LSA_ROUTE_STATS"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lidarslam_amd as L

ctx = L.Context(0)
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx.L.lsa_set_knn_lanes(ctx.h, L.PLANE, lanes)
frames = [L.synth_frame(128, 1000, f)[0] for f in range(3)]
ctx.upload_frame(frames[0]); ctx.extract_keypoints()
ctx.upload_frame(frames[1]); c = ctx.extract_keypoints()
T = np.eye(4); T[0, 3] = 0.45
for pname, mp, cells in (("ego", L.MatchParams.ego_motion(saturation_distance=5.0), (0.5, 0.25)), ("loc", L.MatchParams.localization(saturation_distance=2.0), (0.75, 0.6))):
    for k in (L.EDGE, L.PLANE):
        ctx.set_target_from_set(k, L.SET_RAW_PREVIOUS, cell=cells[k])
    for mask in (1, 2, 3):
        for rep in range(4):
            ctx.match_types(mask, L.SET_RAW_CURRENT, mp, T, slot=L.TARGET_PREVIOUS, histograms=False)
        ctx.sync()
        nbe = (int(c[0]) * 8 + 255) // 256 if mask & 1 else 0
        nbp = (int(c[1]) * lanes + 255) // 256 if mask & 2 else 0
        se, sp = (nbe + 7) // 8, (nbp + 7) // 8
        grid = 8 * (se + sp)
        tr = ctx.match_trace(grid)
        ok = tr[:, 0] > 0
        t0 = tr[ok, 0].min()
        start, mid, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0
        is_edge = (np.arange(grid) // 8) < se
        line = "%s mask %d span %.1f us |" % (pname, mask, end[ok].max())
        for name, m in (("edge", ok & is_edge), ("plane", ok & ~is_edge)):
            if m.sum():
                line += " %s: blocks %d search p50/p90/max %.1f %.1f %.1f model p50/max %.1f %.1f |" % (name, m.sum(), np.median((mid - start)[m]), np.percentile((mid - start)[m], 90), (mid - start)[m].max(), np.median((end - mid)[m]), (end - mid)[m].max())
        print(line)
