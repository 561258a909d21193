#!/usr/bin/env python3
"""Regenerates tests/golden/mini_seq.npz.

The reference ships no golden vectors and cannot be built or run here (C++ with Eigen / PCL / Ceres /
nanoflann, none installed), so these fixtures are produced by the CPU ORACLE (oracle/, the
restatement of the reference's algorithm) -- parity with the reference itself stays UNPINNED.  They
pin the oracle against regressions and give the GPU path a second, file-based target.

Contents (miniature sensor: 8 rings x 400 firings, 4 consecutive frames of sequence seed 1000):
  frame{f}            input scan, POINT_DTYPE (32-byte LidarPoint)
  stamp{f}            header stamp [us]
  dbg{f}              10 x N float32: SpinningSensorKeypointExtractor::GetDebugArray in id order
  kp{f}_{k}           keypoints of type k
  ego_status/_weights/_records{k}   KeypointsMatcher results, frame 1 on frame 0, ego-motion setup
  loc_status/_weights/_records{k}   same, localization setup (RANSAC line neighbours)
  acc_w, acc_cost, acc_g, acc_H     normal equations of the ego-motion records at acc_w
  lm_pose, lm_summary               LocalOptimizer::Solve on them
  undist                            RefineUndistortion of kp1_1 between two poses
  poses                             Slam::GetWorldTransform after each of the 4 frames (EgoMotion = 3)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lidarslam_amd import MatchParams, synth_frame  # noqa: E402
from oracle import oracle as O  # noqa: E402

MODEL, SEED, NFRAMES = 8, 1000, 4


def main():
    out = {}
    ex = O.Extractor()
    frames = []
    for f in range(NFRAMES):
        pts, stamp = synth_frame(MODEL, SEED, f)
        frames.append((pts, stamp))
        ex.compute(pts)
        out[f"frame{f}"] = pts
        out[f"stamp{f}"] = np.array([stamp], np.uint64)
        out[f"dbg{f}"] = np.stack([ex.debug(i) for i in range(10)])
        for k in range(3):
            out[f"kp{f}_{k}"] = ex.keypoints(k)
    out["azimuthal_resolution"] = np.array([ex.azimuthal_resolution], np.float32)

    pose = np.eye(4)
    pose[0, 3] = 0.45
    pose[:3, :3] = [[np.cos(0.01), -np.sin(0.01), 0], [np.sin(0.01), np.cos(0.01), 0], [0, 0, 1]]
    out["match_pose"] = pose
    ego = MatchParams.ego_motion(saturation_distance=5.0)
    loc = MatchParams.localization(saturation_distance=2.0)
    recs, stats = [], []
    for k in range(3):
        cur, tgt = out[f"kp1_{k}"], out[f"kp0_{k}"]
        st, w, rec, hist = O.match(cur, tgt, k, ego, pose)
        out[f"ego_status{k}"], out[f"ego_weights{k}"], out[f"ego_records{k}"], out[f"ego_hist{k}"] = st, w, rec, hist
        if k < 2:
            recs.append(rec)
            stats.append(st)
        st, w, rec, hist = O.match(cur, tgt, k, loc, pose)
        out[f"loc_status{k}"], out[f"loc_weights{k}"], out[f"loc_records{k}"], out[f"loc_hist{k}"] = st, w, rec, hist
    rec, st = np.concatenate(recs), np.concatenate(stats)
    w6 = np.array([0.45, 0.01, -0.02, 0.001, -0.002, 0.012])
    cost, g, H, nv = O.accumulate(rec, st, 5.0, w6)
    out["acc_w"], out["acc_cost"], out["acc_g"], out["acc_H"], out["acc_nvalid"] = w6, np.array([cost]), g, H, np.array([nv])
    lm_pose, lm_w, summ, costs = O.lm_solve(rec, st, 5.0, pose)
    out["lm_pose"], out["lm_summary"], out["lm_costs"] = lm_pose, summ, costs

    H0 = np.eye(4)
    H1 = np.eye(4)
    H1[:3, 3] = [0.5, 0.02, -0.01]
    c, s = np.cos(0.02), np.sin(0.02)
    H1[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    out["undist_H0"], out["undist_H1"] = H0, H1
    out["undist"] = O.undistort(out["kp1_1"], H0, H1, -0.1, 0.0)

    s = O.Slam(EgoMotion=3)
    poses = []
    for f, (pts, stamp) in enumerate(frames):
        s.add_frame(pts, stamp, f)
        poses.append(s.world_transform())
    out["poses"] = np.stack(poses)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mini_seq.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays")


if __name__ == "__main__":
    main()
