"""First-light check on a GPU box: kernel-level parity vs the oracle on one VLP-16 scan pair + short pipeline run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lidarslam_amd as L
from lidarslam_amd import Context, Slam, ExtractParams, MatchParams, synth_frame, synth_pose
from oracle import oracle as O

model = int(os.environ.get("MODEL", "16"))
ctx = Context(0)
p0, s0 = synth_frame(model, 1000, 0)
p1, s1 = synth_frame(model, 1000, 1)
ex = O.Extractor()
ok = True
for f, pts in enumerate([p0, p1]):
    ctx.upload_frame(pts)
    t = time.time(); c = ctx.extract_keypoints(); dt = time.time() - t
    co = ex.compute(pts)
    print(f"frame {f}: gpu counts {c} oracle {co}  azres {ctx.azimuthal_resolution} / {ex.azimuthal_resolution}  {dt*1e3:.2f} ms")
    for i, nm in enumerate(O.DEBUG_NAMES):
        a, b = ctx.debug_array(i), ex.debug(i)
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        if not same:
            d = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32))
            print(f"   {nm}: {d.size} mismatches, first {d[:5]} gpu {a[d[:5]]} orc {b[d[:5]]}")
            ok = False
    for k in range(3):
        a, b = ctx.keypoints(L.SET_RAW_CURRENT, k), ex.keypoints(k)
        if a.tobytes() != b.tobytes():
            print(f"   keypoints type {k} differ: {a.size} vs {b.size}")
            ok = False
print("EXTRACT PARITY", ok)

# matching: current = frame1 keypoints vs target = frame0 keypoints (device resident previous set)
pose = np.eye(4); pose[0, 3] = 0.45
for k, mp in ((0, MatchParams.ego_motion(saturation_distance=5.0)), (1, MatchParams.ego_motion(saturation_distance=5.0))):
    ctx.set_target_from_set(k, L.SET_RAW_PREVIOUS)
    hist = ctx.match(k, L.SET_RAW_CURRENT, mp, pose, slot=L.TARGET_PREVIOUS)
    st, w, rec = ctx.match_results(k, L.SET_RAW_CURRENT)
    cur, tgt = ctx.keypoints(L.SET_RAW_CURRENT, k), ctx.keypoints(L.SET_RAW_PREVIOUS, k)
    so, wo, ro, ho = O.match(cur, tgt, k, mp, pose)
    print(f"match type {k}: hist gpu {hist} orc {ho} slow {ctx.slow_queries()}")
    print("   status equal", np.array_equal(st, so), " weights equal", np.array_equal(w, wo), " records equal", np.array_equal(rec, ro),
          " max|drec|", float(np.abs(rec - ro).max()) if rec.size else 0)
    ok &= np.array_equal(st, so) and np.array_equal(rec, ro)
# localization style (ransac edges)
for k, mp in ((0, MatchParams.localization(saturation_distance=2.0)),):
    hist = ctx.match(k, L.SET_RAW_CURRENT, mp, pose, slot=L.TARGET_PREVIOUS)
    st, w, rec = ctx.match_results(k, L.SET_RAW_CURRENT)
    cur, tgt = ctx.keypoints(L.SET_RAW_CURRENT, k), ctx.keypoints(L.SET_RAW_PREVIOUS, k)
    so, wo, ro, ho = O.match(cur, tgt, k, mp, pose)
    print(f"match(ransac) type {k}: hist gpu {hist} orc {ho}")
    print("   status equal", np.array_equal(st, so), " records equal", np.array_equal(rec, ro))
    ok &= np.array_equal(st, so) and np.array_equal(rec, ro)
# accumulate: plane records
mp = MatchParams.ego_motion(saturation_distance=5.0)
ctx.match(0, L.SET_RAW_CURRENT, mp, pose, slot=L.TARGET_PREVIOUS); ctx.match(1, L.SET_RAW_CURRENT, mp, pose, slot=L.TARGET_PREVIOUS)
w6 = np.array([0.45, 0.01, -0.02, 0.001, -0.002, 0.003])
cg, gg, Hg, ng = ctx.accumulate(3, w6)
tot = None
for k in (0, 1):
    st, w, rec = ctx.match_results(k, L.SET_RAW_CURRENT)
    co_, go, Ho, no = O.accumulate(rec, st, 5.0, w6)
    tot = (co_, go, Ho, no) if tot is None else (tot[0] + co_, tot[1] + go, tot[2] + Ho, tot[3] + no)
print("accumulate: cost", cg, tot[0], "nvalid", ng, tot[3], "rel err g", np.abs(gg - tot[1]).max() / np.abs(tot[1]).max(),
      "rel err H", np.abs(Hg - tot[2]).max() / np.abs(tot[2]).max())

# pipeline
sg, so_ = Slam(0, EgoMotion=3), O.Slam(EgoMotion=3)
for f in range(6):
    pts, stamp = synth_frame(model, 1000, f)
    t = time.time(); sg.add_frame(pts, stamp, f); dtg = time.time() - t
    t = time.time(); so_.add_frame(pts, stamp, f); dto = time.time() - t
    Tg, To = sg.world_transform(), so_.world_transform()
    D = np.linalg.inv(To) @ Tg
    print(f"frame {f}: gpu {dtg*1e3:.1f} ms cpu {dto*1e3:.1f} ms  dpos {np.linalg.norm(D[:3,3]):.2e} drot {np.arccos(min(1,(np.trace(D[:3,:3])-1)/2)):.2e}  pos {Tg[:3,3].round(4)}")
print("ALL OK" if ok else "MISMATCH")
