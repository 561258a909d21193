"""Micro-benchmark of the one-launch match (k_search_all) on the inputs of a real VLS-128 frame.

  python scripts/match_microbench.py dump  FILE      run the pipeline (tree library) for 30 frames, save what its last
                                                     ego-motion and localization matches saw
  [LSA_LIB=_variants/lib_x.so] python scripts/match_microbench.py run FILE [reps]
                                                     replay both matches `reps` times back to back, print us / launch
Lets library variants (scripts/build_variant.sh), also ones that compute nonsense on purpose, be timed on identical
inputs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lidarslam_amd as L

mode, path = sys.argv[1], sys.argv[2]
if mode == "dump":
    slam = L.Slam(0, EgoMotion=3)
    n = 30
    for f in range(n):
        pts, stamp = L.synth_frame(128, 1000, f)
        slam.add_frame(pts, stamp, f)
    ctx = slam.context()
    ctx.sync()
    out = {"pose": np.asarray(slam.world_transform())}
    for k in range(2):
        out[f"sub{k}"] = slam.target_submap(k)
        out[f"work{k}"] = slam.keypoints(k, 0)
        out[f"raw{k}"] = ctx.keypoints(L.SET_RAW_CURRENT, k)
        out[f"prev{k}"] = ctx.keypoints(L.SET_RAW_PREVIOUS, k)
    out["leaf"] = np.array([slam.get_param("KnnCellScaleMapsEdges") * 0.3, slam.get_param("KnnCellScaleMaps") * 0.6, slam.get_param("KnnCellSizeEgoMotionEdges"), slam.get_param("KnnCellSizeEgoMotion")])
    np.savez(path, **out)
    print("saved", {k: v.shape for k, v in out.items()})
    sys.exit(0)

reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
with np.load(path) as z:
    d = {k: z[k] for k in z.files}
ctx = L.Context(0)
for k in range(2):
    ctx.set_target(k, d[f"sub{k}"], cell=float(d["leaf"][k]), slot=L.TARGET_MAP)
    ctx.set_target(k, d[f"prev{k}"], cell=float(d["leaf"][2 + k]), slot=L.TARGET_PREVIOUS)
    ctx.set_keypoints(L.SET_WORKING, k, d[f"work{k}"])
    ctx.set_keypoints(L.SET_RAW_CURRENT, k, d[f"raw{k}"])
rel = np.eye(4); rel[0, 3] = 0.5
legs = (("loc", L.MatchParams.localization(saturation_distance=1.0), L.SET_WORKING, d["pose"], L.TARGET_MAP),
        ("ego", L.MatchParams.ego_motion(saturation_distance=3.0), L.SET_RAW_CURRENT, rel, L.TARGET_PREVIOUS))
res = []
for name, mp, qs, pose, slot in legs:
    h = ctx.match_types(3, qs, mp, pose, slot=slot)  # warm-up (+ the histograms: a checksum of what the variant computes)
    best = 1e9
    for _ in range(5):
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.match_types(3, qs, mp, pose, slot=slot, histograms=False)
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    res.append("%s %.1f us (ok %d/%d)" % (name, 1e6 * best, int(h[0][0] + h[1][0]), int(h[:2].sum())))
print(os.environ.get("LSA_LIB", "tree"), " | ".join(res))
