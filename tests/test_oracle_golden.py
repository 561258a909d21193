"""The CPU oracle against the committed golden vectors (tests/golden/mini_seq.npz, produced by
tests/golden/make_golden.py).  The reference holds no fixtures for this path, so
these vectors come from the oracle itself: they pin it against regressions; parity with the
reference stays unpinned."""
import numpy as np

from conftest import bits, pose_diff


def test_extractor_matches_golden(O, golden):
    ex = O.Extractor()
    for f in range(4):
        counts = ex.compute(golden[f"frame{f}"])
        for i, name in enumerate(O.DEBUG_NAMES):
            assert np.array_equal(bits(ex.debug(i)), bits(golden[f"dbg{f}"][i])), (f, name)
        for k in range(3):
            assert counts[k] == golden[f"kp{f}_{k}"].size
            assert ex.keypoints(k).tobytes() == golden[f"kp{f}_{k}"].tobytes()
    assert np.float32(ex.azimuthal_resolution) == golden["azimuthal_resolution"][0]


def test_matcher_matches_golden(O, L, golden):
    pose = golden["match_pose"]
    for tag, mp in (("ego", L.MatchParams.ego_motion(saturation_distance=5.0)), ("loc", L.MatchParams.localization(saturation_distance=2.0))):
        for k in range(3):
            st, w, rec, hist = O.match(golden[f"kp1_{k}"], golden[f"kp0_{k}"], k, mp, pose)
            assert np.array_equal(st, golden[f"{tag}_status{k}"]), (tag, k)
            assert np.array_equal(bits(w), bits(golden[f"{tag}_weights{k}"]))
            assert np.array_equal(bits(rec), bits(golden[f"{tag}_records{k}"]))
            assert np.array_equal(hist, golden[f"{tag}_hist{k}"])


def test_normal_equations_and_lm_match_golden(O, golden):
    rec = np.concatenate([golden["ego_records0"], golden["ego_records1"]])
    st = np.concatenate([golden["ego_status0"], golden["ego_status1"]])
    cost, g, H, nv = O.accumulate(rec, st, 5.0, golden["acc_w"])
    assert cost == golden["acc_cost"][0] and nv == golden["acc_nvalid"][0]
    assert np.array_equal(g, golden["acc_g"]) and np.array_equal(H, golden["acc_H"])
    pose, w, summ, costs = O.lm_solve(rec, st, 5.0, golden["match_pose"])
    assert np.array_equal(summ, golden["lm_summary"])
    assert np.allclose(pose, golden["lm_pose"], atol=1e-13, rtol=0)


def test_undistortion_matches_golden(O, golden):
    out = O.undistort(golden["kp1_1"], golden["undist_H0"], golden["undist_H1"], -0.1, 0.0)
    assert out.tobytes() == golden["undist"].tobytes()


def test_pipeline_poses_match_golden(O, golden):
    s = O.Slam(EgoMotion=3)
    for f in range(4):
        s.add_frame(golden[f"frame{f}"], int(golden[f"stamp{f}"][0]), f)
        dp, da = pose_diff(golden["poses"][f], s.world_transform())
        assert dp < 1e-10 and da < 1e-7, (f, dp, da)
