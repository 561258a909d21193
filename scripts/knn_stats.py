"""How many queries does each stage of the kNN cascade see?  (VLS-128 frame pair, ego-motion and map-like targets)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lidarslam_amd as L

lib = L.lib()
ctx = L.Context(0)
frames = [L.synth_frame(128, 1000, f)[0] for f in range(12)]
ctx.upload_frame(frames[0]); ctx.extract_keypoints()
ctx.upload_frame(frames[1]); c = ctx.extract_keypoints()
T = np.eye(4); T[0, 3] = 0.45
for name, mp in (("ego", L.MatchParams.ego_motion(saturation_distance=5.0)), ("loc", L.MatchParams.localization(saturation_distance=2.0))):
    for k, cell in ((L.EDGE, 0.5), (L.PLANE, 0.25)):
        ctx.set_target_from_set(k, L.SET_RAW_PREVIOUS, cell=cell)
        hist = ctx.match(k, L.SET_RAW_CURRENT, mp, T, slot=L.TARGET_PREVIOUS)
        print(name, "type", k, "queries", int(c[k]), "second stage", ctx.slow_queries(), "exhaustive", lib.lsa_match_exhaustive_queries(ctx.h), "hist", hist.tolist())
