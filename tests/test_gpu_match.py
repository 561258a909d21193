"""GPU parity, seams 2 and 3: KeypointsMatcher::BuildMatchResiduals (exact kNN + model fit) and the
LocalOptimizer evaluation / solve through the C ABI against the CPU oracle.
Bar: match status, weights and residual records bit-exact (integer / decision work and the float and
double arithmetic behind it); normal equations within 1e-12 relative (the GPU tree reduction sums in
another order than the reference's sequential loop); solved poses within 1e-9 (north star: 1e-4)."""
import time

import numpy as np
import pytest

from conftest import bits, pose_diff

pytestmark = pytest.mark.gpu


def perturbed(dx=0.45, yaw=0.01, dz=0.0):
    T = np.eye(4)
    c, s = np.cos(yaw), np.sin(yaw)
    T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    T[:3, 3] = [dx, 0.02, dz]
    return T


def se3(dx, dy, dz, rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    T = np.eye(4)
    T[:3, :3] = [[cy * cz, sx * sy * cz - cx * sz, cx * sy * cz + sx * sz], [cy * sz, sx * sy * sz + cx * cz, cx * sy * sz - sx * cz], [-sy, sx * cy, cx * cy]]
    T[:3, 3] = [dx, dy, dz]
    return T


def assert_match_equal(ctx, O, L, k, cur, tgt, mp, pose, cell=None):
    ctx.set_keypoints(L.SET_WORKING, k, cur)
    ctx.set_target(k, tgt, cell=cell)
    hist = ctx.match(k, L.SET_WORKING, mp, pose)
    st, w, rec = ctx.match_results(k, L.SET_WORKING)
    so, wo, ro, ho = O.match(cur, tgt, k, mp, pose)
    assert st.size == so.size == cur.size
    bad = np.flatnonzero(st != so)
    assert bad.size == 0, f"status differs at {bad[:8]}: gpu {st[bad[:8]]} oracle {so[bad[:8]]}"
    assert hist.tolist() == ho.tolist()
    assert np.array_equal(bits(w), bits(wo)), "weights"
    assert np.array_equal(bits(rec), bits(ro)), f"records: max abs diff {np.abs(rec - ro).max()}"
    return st


@pytest.fixture(scope="module")
def kps(O, L):
    """keypoints of two consecutive scans per sensor model, extracted by the oracle"""
    out = {}
    for model in (8, 16, 64, 128):
        ex = O.Extractor()
        per = []
        for f in range(2):
            pts, _ = L.synth_frame(model, 1000, f)
            ex.compute(pts)
            per.append([ex.keypoints(k) for k in range(3)])
        out[model] = per
    return out


@pytest.mark.parametrize("model", [8, 16, 64, 128])  # fixtures, VLP-16, HDL-64 (BASELINE config 3), VLS-128
def test_ego_motion_matching_bit_exact(gpu_ctx, O, L, kps, model):
    """per-ring edge neighbourhoods + plane fits, current scan on the previous scan (Slam.cxx:877-911)"""
    prev, cur = kps[model]
    mp = L.MatchParams.ego_motion(saturation_distance=5.0)
    for k in (L.EDGE, L.PLANE):
        st = assert_match_equal(gpu_ctx, O, L, k, cur[k], prev[k], mp, perturbed())
        assert (st == 0).sum() > 20


@pytest.mark.parametrize("model", [8, 16, 64, 128])
def test_localization_matching_bit_exact(gpu_ctx, O, L, kps, model):
    """RANSAC line neighbourhoods, planes and blobs (Slam.cxx:1055-1090)"""
    prev, cur = kps[model]
    mp = L.MatchParams.localization(saturation_distance=2.0)
    for k in (L.EDGE, L.PLANE, L.BLOB):
        n = 20000 if k == L.BLOB else None  # blobs: every third point, keep the oracle run short
        assert_match_equal(gpu_ctx, O, L, k, cur[k][:n], prev[k], mp, perturbed(), cell=1.2)


@pytest.mark.parametrize("edge_k,plane_k,blob_k", [(2, 3, 4), (5, 4, 6), (9, 7, 9), (12, 8, 12), (16, 16, 16)])
def test_matching_with_other_neighbour_counts(gpu_ctx, O, L, kps, edge_k, plane_k, blob_k):
    prev, cur = kps[16]
    for single in (0, 1):
        mp = L.MatchParams(single_edge_per_ring=single, edge_nb_neighbors=edge_k, edge_min_nb_neighbors=2, plane_nb_neighbors=plane_k,
                           blob_nb_neighbors=blob_k, max_neighbors_distance=3.0, edge_max_model_error=0.1, plane_max_model_error=0.1)
        for k in range(3):
            assert_match_equal(gpu_ctx, O, L, k, cur[k][:3000], prev[k], mp, perturbed(0.3, -0.02, 0.05))


@pytest.mark.parametrize("cell", [0.3, 1.0, 4.0, 50.0])
def test_knn_is_exact_for_any_grid_resolution(gpu_ctx, O, L, kps, cell):
    """the search grid is an implementation detail: results must not depend on the cell size"""
    prev, cur = kps[16]
    mp = L.MatchParams.localization()
    assert_match_equal(gpu_ctx, O, L, L.PLANE, cur[1], prev[1], mp, perturbed(), cell=cell)
    assert_match_equal(gpu_ctx, O, L, L.EDGE, cur[0], prev[0], mp, perturbed(), cell=cell)


def test_matching_edge_cases(gpu_ctx, O, L, kps):
    prev, cur = kps[16]
    ego = L.MatchParams.ego_motion()
    # empty target: every keypoint stays UNKOWN, histogram empty (KeypointsMatcher.cxx:53-58)
    st = assert_match_equal(gpu_ctx, O, L, 1, cur[1], prev[1][:0], ego, np.eye(4))
    assert np.all(st == 7)
    # fewer target points than neighbours requested
    st = assert_match_equal(gpu_ctx, O, L, 1, cur[1], prev[1][:3], ego, np.eye(4))
    assert np.all(st == 2)
    st = assert_match_equal(gpu_ctx, O, L, 0, cur[0], prev[0][:1], ego, np.eye(4))
    # queries far outside the target's bounding box, sparse target (exhaustive fall-back of the search)
    far = perturbed(400.0, 0.3, 30.0)
    st = assert_match_equal(gpu_ctx, O, L, 1, cur[1], prev[1], ego, far)
    assert np.all(st == 3)
    assert_match_equal(gpu_ctx, O, L, 0, cur[0], prev[0][::40], L.MatchParams.localization(), perturbed())
    assert_match_equal(gpu_ctx, O, L, 0, cur[0], prev[0][::40], ego, perturbed())
    # bad parametrisation
    st = assert_match_equal(gpu_ctx, O, L, 1, cur[1], prev[1], L.MatchParams(plane_nb_neighbors=2), np.eye(4))
    assert np.all(st == 1)
    st = assert_match_equal(gpu_ctx, O, L, 2, cur[2][:500], prev[2], L.MatchParams(blob_nb_neighbors=3), np.eye(4))
    assert np.all(st == 1)
    # no queries at all
    gpu_ctx.set_keypoints(L.SET_WORKING, 1, cur[1][:0])
    assert gpu_ctx.match(1, L.SET_WORKING, ego, np.eye(4)).sum() == 0
    # duplicated target points: exact distance ties are ordered by index
    dup = np.concatenate([prev[1], prev[1]])
    assert_match_equal(gpu_ctx, O, L, 1, cur[1], dup, ego, perturbed())
    # collinear / coincident neighbourhoods (degenerate covariance)
    line = prev[1][:200].copy()
    line["y"], line["z"] = 0, 0
    line["x"] = np.linspace(-5, 5, 200, dtype=np.float32)
    assert_match_equal(gpu_ctx, O, L, 1, cur[1][:500], line, ego, np.eye(4))
    same = np.repeat(prev[1][:1], 50)
    assert_match_equal(gpu_ctx, O, L, 1, cur[1][:500], same, ego, np.eye(4))
    assert_match_equal(gpu_ctx, O, L, 0, cur[0][:500], same, L.MatchParams.localization(), np.eye(4))


def test_device_resident_target_equals_host_target(gpu_ctx, O, L):
    """ego-motion registers on the previous scan's keypoints without leaving the device"""
    ex = O.Extractor()
    gpu_ctx.azimuthal_resolution = 0.0
    for f in range(2):
        pts, _ = L.synth_frame(16, 1000, f)
        gpu_ctx.upload_frame(pts)
        gpu_ctx.extract_keypoints()
        ex.compute(pts)
    mp = L.MatchParams.ego_motion(saturation_distance=5.0)
    for k in (0, 1):
        gpu_ctx.set_target_from_set(k, L.SET_RAW_PREVIOUS)
        gpu_ctx.match(k, L.SET_RAW_CURRENT, mp, perturbed(), slot=L.TARGET_PREVIOUS)
        st, w, rec = gpu_ctx.match_results(k, L.SET_RAW_CURRENT)
        so, wo, ro, _ = O.match(gpu_ctx.keypoints(L.SET_RAW_CURRENT, k), gpu_ctx.keypoints(L.SET_RAW_PREVIOUS, k), k, mp, perturbed())
        assert np.array_equal(st, so) and np.array_equal(bits(rec), bits(ro))


# ---------------------------------------------------------------------------------- seam 3
def setup_residuals(ctx, O, L, kps, model=16, sat=5.0):
    prev, cur = kps[model]
    mp = L.MatchParams.ego_motion(saturation_distance=sat)
    recs, sts = [], []
    for k in (0, 1):
        ctx.set_keypoints(L.SET_WORKING, k, cur[k])
        ctx.set_target(k, prev[k])
        ctx.match(k, L.SET_WORKING, mp, perturbed())
        st, w, rec = ctx.match_results(k, L.SET_WORKING)
        recs.append(rec)
        sts.append(st)
    ctx.set_keypoints(L.SET_WORKING, 2, cur[2][:0])
    ctx.match(2, L.SET_WORKING, mp, perturbed())
    return np.concatenate(recs), np.concatenate(sts)


@pytest.mark.parametrize("model", [16, 64, 128])  # VLP-16, HDL-64 (BASELINE config 3), VLS-128
def test_normal_equations_match_the_oracle(gpu_ctx, O, L, kps, model):
    rec, st = setup_residuals(gpu_ctx, O, L, kps, model)
    for w6 in (np.zeros(6), np.array([0.45, 0.01, -0.02, 0.001, -0.002, 0.012]), np.array([3.0, -2.0, 1.0, 0.3, -0.2, 1.5])):
        cost, g, H, nv = gpu_ctx.accumulate(7, w6)
        co, go, Ho, no = O.accumulate(rec, st, 5.0, w6)
        assert nv == no
        assert abs(cost - co) <= 1e-12 * abs(co)
        assert np.abs(g - go).max() <= 1e-11 * np.abs(go).max()
        assert np.abs(H - Ho).max() <= 1e-12 * np.abs(Ho).max()
        assert np.array_equal(H, H.T)
        c2, _, _, _ = gpu_ctx.accumulate(7, w6, jac=False)
        assert c2 == cost  # the cost does not depend on whether the Jacobian is asked for


def test_normal_equations_are_reproducible_and_additive(gpu_ctx, O, L, kps):
    setup_residuals(gpu_ctx, O, L, kps)
    w6 = np.array([0.4, 0.0, 0.01, 0.0, 0.001, 0.01])
    a = gpu_ctx.accumulate(3, w6)
    b = gpu_ctx.accumulate(3, w6)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])  # fixed-order reduction: bitwise
    e, p = gpu_ctx.accumulate(1, w6), gpu_ctx.accumulate(2, w6)  # residual blocks of the two types are independent
    assert abs(e[0] + p[0] - a[0]) <= 1e-12 * a[0] and e[3] + p[3] == a[3]
    assert np.abs(e[2] + p[2] - a[2]).max() <= 1e-12 * np.abs(a[2]).max()


@pytest.mark.parametrize("two_d", [False, True])
def test_lm_solve_matches_the_oracle(gpu_ctx, O, L, kps, two_d):
    rec, st = setup_residuals(gpu_ctx, O, L, kps)
    prior = perturbed(0.3, 0.0)
    pose, summ, costs = gpu_ctx.solve(7, prior, max_iter=15, two_d=two_d)
    po, wo, so, co = O.lm_solve(rec, st, 5.0, prior, max_iter=15, two_d=two_d)
    dp, da = pose_diff(po, pose)
    assert dp < 1e-9 and da < 1e-7, (dp, da)
    assert summ[0] == so[0] and summ[2] == so[2]  # accepted steps and iterations; evaluations differ by design
    assert abs(costs[1] - co[1]) <= 1e-9 * co[1] and costs[1] <= costs[0]
    # solving again from the optimum: "num_successful_steps == 1", the ICP stop criterion
    pose2, summ2, _ = gpu_ctx.solve(7, pose, max_iter=15, two_d=two_d)
    assert summ2[0] == 1
    cov, err = gpu_ctx.registration_error(7, pose, two_d=two_d)
    covo, erro = O.covariance(rec, st, 5.0, pose) if not two_d else (None, None)
    if covo is not None:
        assert np.abs(cov - covo).max() <= 1e-6 * np.abs(covo).max() and np.allclose(err, erro, rtol=1e-6)


@pytest.mark.parametrize("model", [16, 128])
def test_concurrent_types_equal_sequential_calls(gpu_ctx, O, L, kps, model):
    """lsa_match_types (one ICP iteration's matching step, types side by side on the device) leaves the
    records, status and histograms of three lsa_match calls; the asynchronous form (no histogram) reports
    the number of matches through lsa_accumulate."""
    prev, cur = kps[model]
    mp = L.MatchParams.localization(saturation_distance=2.0)
    pose = perturbed()
    for k in (L.EDGE, L.PLANE, L.BLOB):
        n = 20000 if k == L.BLOB else None
        gpu_ctx.set_keypoints(L.SET_WORKING, k, cur[k][:n])
        gpu_ctx.set_target(k, prev[k], cell=1.2)
    seq = {}
    for k in (L.EDGE, L.PLANE, L.BLOB):
        hist = gpu_ctx.match(k, L.SET_WORKING, mp, pose)
        seq[k] = (hist,) + gpu_ctx.match_results(k, L.SET_WORKING)
    ref_acc = gpu_ctx.accumulate(7, np.zeros(6))
    hists = gpu_ctx.match_types(7, L.SET_WORKING, mp, pose)
    for k in (L.EDGE, L.PLANE, L.BLOB):
        st, w, rec = gpu_ctx.match_results(k, L.SET_WORKING)
        assert hists[k].tolist() == seq[k][0].tolist()
        assert np.array_equal(st, seq[k][1]) and np.array_equal(bits(w), bits(seq[k][2])) and np.array_equal(bits(rec), bits(seq[k][3]))
    # a subset of the types, asynchronously: the other type's records stay, the count comes with the evaluation
    assert gpu_ctx.match_types((1 << L.EDGE) | (1 << L.PLANE), L.SET_WORKING, mp, pose, histograms=False) is None
    acc = gpu_ctx.accumulate(7, np.zeros(6))
    assert acc[3] == ref_acc[3] == sum(int(seq[k][0][0]) for k in seq)
    assert acc[0] == ref_acc[0] and np.array_equal(acc[2], ref_acc[2])


def test_histograms_of_earlier_matches_stay_readable(gpu_ctx, O, L, kps):
    """lsa_match_serial / lsa_match_histogram (MatchingResults::NbMatches without a read-back on the critical path):
    the histogram of a match is still there after later matches of the same type, up to 128 of them (half of the ring:
    the other half may be being cleared), across the ring's wrap-around"""
    prev, cur = kps[16]
    pose = perturbed()
    for k in (L.EDGE, L.PLANE):
        gpu_ctx.set_keypoints(L.SET_WORKING, k, cur[k])
        gpu_ctx.set_target(k, prev[k], cell=1.0)
    seen = []
    for i in range(300):
        mp = L.MatchParams.localization(saturation_distance=2.0 - 0.05 * (i % 20))
        mp.max_neighbors_distance = 5.0 - 0.2 * (i % 20)  # a different histogram every time
        hists = gpu_ctx.match_types(3, L.SET_WORKING, mp, pose, histograms=(i % 7 == 0 or i >= 150))
        seen.append((gpu_ctx.match_serial(L.EDGE), gpu_ctx.match_serial(L.PLANE), None if hists is None else hists.copy()))
    assert seen[-1][0] == seen[0][0] + 299
    checked = 0
    for se, sp, h in seen[-128:]:
        if h is None:
            continue
        assert gpu_ctx.match_histogram(L.EDGE, se).tolist() == h[L.EDGE].tolist()
        assert gpu_ctx.match_histogram(L.PLANE, sp).tolist() == h[L.PLANE].tolist()
        checked += 1
    assert checked == 128
    assert len({tuple(h[L.PLANE].tolist()) for _, _, h in seen if h is not None}) > 5
    with pytest.raises(L.LsaError):
        gpu_ctx.match_histogram(L.EDGE, seen[-129][0])  # 128 matches ago: gone
    with pytest.raises(L.LsaError):
        gpu_ctx.match_histogram(L.EDGE, seen[-1][0] + 1)  # not enqueued yet


def test_interpolated_bounding_boxes_are_those_of_the_undistorted_keypoints(gpu_ctx, O, L, kps):
    """lsa_keypoint_bboxes_begin_interp: the box of the keypoints under the pose interpolated at every point's own
    time = the box of the same keypoints after lsa_undistort with that motion (what the sub-map prediction relies on)"""
    _, cur = kps[16]
    for k in (L.EDGE, L.PLANE, L.BLOB):
        gpu_ctx.set_keypoints(L.SET_RAW_CURRENT, k, cur[k][:5000])
    t0, t1 = gpu_ctx.keypoint_time_range(L.SET_RAW_CURRENT)
    alltimes = np.concatenate([cur[k][:5000]["time"] for k in range(3)])
    assert (t0, t1) == (alltimes.min(), alltimes.max()) and t1 - t0 > 0.05
    H0, H1 = perturbed(0.4, 0.05), perturbed(-0.3, -0.04)
    mn, mx = gpu_ctx.keypoint_bboxes(L.SET_RAW_CURRENT, H0, H1, t0, t1)
    for k in (L.EDGE, L.PLANE, L.BLOB):
        und = O.undistort(cur[k][:5000], H0, H1, t0, t1)
        for d, f in enumerate("xyz"):
            assert abs(mn[k, d] - und[f].min()) < 1e-5 and abs(mx[k, d] - und[f].max()) < 1e-5
    # the rigid form, for comparison: the motion between H0 and H1 moves the boxes
    rn, rx = gpu_ctx.keypoint_bboxes(L.SET_RAW_CURRENT, H0)
    assert np.abs(rn - mn).max() > 0.05


def test_normal_equations_arrive_through_the_mailbox(gpu_ctx, O, L, kps):
    """the fast hand-over of lsa_accumulate (partial sums written straight into coherent host memory as 8-byte
    {tag, half} granules, one atomic store each) is in use, and what arrives through it is bit for bit what the device
    folds from its own per-block partials (LSA_MAILBOX_CHECK makes every evaluation do that comparison itself and
    fail on a difference)"""
    import os

    assert L.lib().lsa_mailbox_active(gpu_ctx.h) == 1
    os.environ["LSA_MAILBOX_CHECK"] = "1"
    try:
        ctx = L.Context(0)
    finally:
        del os.environ["LSA_MAILBOX_CHECK"]
    setup_residuals(ctx, O, L, kps, 128)
    setup_residuals(gpu_ctx, O, L, kps, 128)
    rng = np.random.default_rng(5)
    for i in range(200):
        w6 = np.array([0.45, 0.01, -0.02, 0.001, -0.002, 0.012]) + 0.01 * rng.standard_normal(6)
        a = ctx.accumulate(7, w6)  # raises if mailbox and device fold disagree
        if i % 20 == 0:
            b = gpu_ctx.accumulate(7, w6)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
    # a whole pipeline under the check
    os.environ["LSA_MAILBOX_CHECK"] = "1"
    try:
        s = L.Slam(0, EgoMotion=3, DeviceLM=0)
    finally:
        del os.environ["LSA_MAILBOX_CHECK"]
    for f in range(4):
        pts, stamp = L.synth_frame(16, 1000, f)
        s.add_frame(pts, stamp, f)
    s.close()
    ctx.close()


@pytest.mark.parametrize("two_d", [False, True])
@pytest.mark.parametrize("model", [16, 64, 128])  # VLP-16, HDL-64 (BASELINE config 3), VLS-128
def test_one_launch_solve_equals_the_host_driven_loop(gpu_ctx, O, L, kps, two_d, model):
    """lsa_solve_device: the trust-region loop of LocalOptimizer::Solve inside one kernel (blocks exchanging their
    partial sums through tagged granules) takes the decisions of the host-driven loop and of the oracle"""
    rec, st = setup_residuals(gpu_ctx, O, L, kps, model)
    for dx, yaw in ((0.3, 0.0), (0.45, 0.01), (0.0, 0.0)):
        prior = perturbed(dx, yaw)
        w0 = np.array([dx, 0.02, 0.0, 0.0, 0.0, yaw])
        r = gpu_ctx.solve_device(7, w0, max_iter=15, two_d=two_d)
        pose, summ, costs = gpu_ctx.solve(7, prior, max_iter=15, two_d=two_d)
        po, wo, so, co = O.lm_solve(rec, st, 5.0, prior, max_iter=15, two_d=two_d)
        assert (r.num_successful_steps, r.num_iterations) == (summ[0], summ[2]) == (so[0], so[2])
        assert r.num_evaluations == summ[3] and not r.skipped
        assert np.abs(np.array(r.pose) - wo).max() < 1e-9, (list(r.pose), wo)
        assert abs(r.final_cost - costs[1]) <= 1e-10 * costs[1] and abs(r.initial_cost - costs[0]) <= 1e-10 * costs[0]
        # the normal equations it returns are those at the returned pose
        c, g, H, nv = gpu_ctx.accumulate(7, np.array(r.pose))
        assert nv == r.num_matches and abs(c - r.cost) <= 1e-10 * c
        assert np.abs(np.array(r.H).reshape(6, 6) - H).max() <= 1e-10 * np.abs(H).max()
        assert np.abs(np.array(r.g) - g).max() <= 1e-8 * max(np.abs(g).max(), 1e-3 * np.abs(H).max())
    # bitwise reproducible, launch after launch
    a = gpu_ctx.solve_device(7, w0, max_iter=15, two_d=two_d)
    b = gpu_ctx.solve_device(7, w0, max_iter=15, two_d=two_d)
    assert list(a.pose) == list(b.pose) and a.final_cost == b.final_cost and list(a.H) == list(b.H)
    # fewer matches than asked for: nothing is optimised, the prior comes back (Slam.cxx:919-923, 1098-1107)
    s = gpu_ctx.solve_device(7, w0, max_iter=15, two_d=two_d, min_matches=10 ** 7)
    assert s.skipped == 1 and s.num_evaluations == 1 and list(s.pose) == list(w0) and s.num_matches == a.num_matches
    # iteration cap
    m = gpu_ctx.solve_device(7, np.array([0.45, 0.02, 0, 0, 0, 0.01]), max_iter=1, two_d=two_d)
    assert m.num_iterations == 1 and m.termination in (3, 7, 8)
    assert gpu_ctx.solve_device_fallbacks() == 0


def test_one_launch_solve_with_no_residuals(gpu_ctx, L):
    for k in range(3):
        gpu_ctx.set_keypoints(L.SET_WORKING, k, np.zeros(0, L.POINT_DTYPE))
        gpu_ctx.match(k, L.SET_WORKING, L.MatchParams.localization(saturation_distance=2.0), np.eye(4))
    r = gpu_ctx.solve_device(7, np.zeros(6), min_matches=20)
    assert r.skipped == 1 and r.num_matches == 0
    r = gpu_ctx.solve_device(7, np.zeros(6), min_matches=0)
    assert r.skipped == 0 and r.termination == 2 and r.num_successful_steps == 1


def test_targets_prepared_ahead_give_the_same_matches(O, L, kps):
    """lsa_stage_target_ahead / lsa_drop_target_ahead / lsa_prepare_previous_targets through the C ABI: a target
    uploaded and indexed ahead of time on the look-ahead stream is taken over only when it is the very cloud (and cell
    size) that is asked for, and then gives the matches of a target set the usual way."""
    prev, cur = kps[16]
    mp, pose = L.MatchParams.localization(saturation_distance=2.0), perturbed()
    ctx = L.Context(0)
    adopted = lambda: ctx.L.lsa_staged_targets_adopted(ctx.h)
    ctx.set_keypoints(L.SET_WORKING, L.PLANE, cur[L.PLANE])
    ctx.set_target(L.PLANE, prev[L.PLANE], cell=1.0)
    ref_hist = ctx.match(L.PLANE, L.SET_WORKING, mp, pose)
    ref = ctx.match_results(L.PLANE, L.SET_WORKING)
    # ahead, then the same cloud asked for: adopted
    ctx.stage_target(L.PLANE, prev[L.PLANE], ahead=True, cell=1.0)
    ctx.stage_target(L.PLANE, prev[L.PLANE], ahead=False, cell=1.0)
    assert adopted() == 1
    assert ctx.match(L.PLANE, L.SET_WORKING, mp, pose).tolist() == ref_hist.tolist()
    got = ctx.match_results(L.PLANE, L.SET_WORKING)
    assert all(np.array_equal(bits(a), bits(b)) for a, b in zip(got, ref))
    assert ctx.target(L.PLANE).tobytes() == prev[L.PLANE].tobytes()
    # ahead, then another cell size: built the usual way; ahead, then dropped: not adopted either
    ctx.stage_target(L.PLANE, prev[L.PLANE], ahead=True, cell=1.0)
    ctx.stage_target(L.PLANE, prev[L.PLANE], ahead=False, cell=0.7)
    ctx.stage_target(L.PLANE, prev[L.PLANE][:-10], ahead=True, cell=0.7)
    ctx.drop_target_ahead(L.PLANE)
    ctx.stage_target(L.PLANE, prev[L.PLANE][:-10], ahead=False, cell=0.7)
    assert adopted() == 1 and ctx.target(L.PLANE).size == prev[L.PLANE].size - 10
    # the previous-scan targets of the next frame: current raw keypoints now, previous ones after the shift
    taken = lambda: ctx.L.lsa_prepared_targets_adopted(ctx.h)
    pts, _ = L.synth_frame(16, 1000, 0)
    ctx.upload_frame(pts)
    ctx.extract_keypoints()
    ctx.L.lsa_set_target_cell_size(ctx.h, L.TARGET_PREVIOUS, L.PLANE, 0.25)
    ctx.prepare_previous_targets(1 << L.PLANE)
    first = ctx.keypoints(L.SET_RAW_CURRENT, L.PLANE)
    ctx.upload_frame(L.synth_frame(16, 1000, 1)[0])
    ctx.extract_keypoints()  # the shift: what was current is previous now
    ctx.set_target_from_set(L.PLANE, L.SET_RAW_PREVIOUS, cell=0.25)
    assert taken() == 1 and ctx.target(L.PLANE, L.TARGET_PREVIOUS).tobytes() == first.tobytes()
    ctx.prepare_previous_targets(1 << L.PLANE)
    ctx.set_keypoints(L.SET_RAW_CURRENT, L.PLANE, first[:100])  # the set is rewritten: the target built ahead is stale
    ctx.upload_frame(L.synth_frame(16, 1000, 2)[0])
    ctx.extract_keypoints()
    ctx.set_target_from_set(L.PLANE, L.SET_RAW_PREVIOUS, cell=0.25)
    assert taken() == 1 and ctx.target(L.PLANE, L.TARGET_PREVIOUS).tobytes() == first[:100].tobytes()
    ctx.close()


def test_undistortion_inside_the_search_kernel(L, kps):
    """lsa_match_types_undistorted against lsa_undistort + lsa_match_types: working keypoints, histograms, status, weights
    and records bit for bit -- with the undistortion inside the search kernel (every keypoint reached), and in the cases
    where it has to step aside (a type not asked for, a type with invalid parameters, the staged kernels)"""
    prev, cur = kps[128]
    H0 = se3(0.01, -0.02, 0.0, 0.001, 0.0, -0.003)
    H1 = se3(0.32, 0.03, -0.01, -0.004, 0.002, 0.011)
    bad = L.MatchParams.localization(saturation_distance=2.0)
    bad.plane_nb_neighbors = 2
    cases = [(7, L.MatchParams.localization(saturation_distance=2.0), 1), (3, L.MatchParams.localization(saturation_distance=2.0), 1),
             (7, bad, 1), (7, L.MatchParams.localization(saturation_distance=1.0), 0), (7, L.MatchParams.localization(saturation_distance=1.0), 2)]
    a, b = L.Context(0), L.Context(0)
    for mask, mp, fused in cases:
        for ctx in (a, b):
            ctx.set_fused_match(fused)
            for k in (L.EDGE, L.PLANE, L.BLOB):
                ctx.set_keypoints(L.SET_WORKING, k, cur[k][:20000])
                ctx.set_target(k, prev[k], cell=(0.75, 0.6, 0.3)[k])
        ha = a.match_types_undistorted(mask, mp, perturbed(), H0, H1, -0.1, 0.0)
        b.undistort(H0, H1, -0.1, 0.0)
        hb = b.match_types(mask, L.SET_WORKING, mp, perturbed())
        assert ha.tolist() == hb.tolist()
        for k in (L.EDGE, L.PLANE, L.BLOB):
            assert a.keypoints(L.SET_WORKING, k).tobytes() == b.keypoints(L.SET_WORKING, k).tobytes(), (mask, fused, k)
            if (mask >> k) & 1:
                sa, wa, ra = a.match_results(k, L.SET_WORKING)
                sb, wb, rb = b.match_results(k, L.SET_WORKING)
                assert np.array_equal(sa, sb) and np.array_equal(bits(wa), bits(wb)) and np.array_equal(bits(ra), bits(rb))
    a.close()
    b.close()


@pytest.mark.parametrize("model", [16, 64, 128])  # VLP-16, HDL-64 (BASELINE config 3), VLS-128
def test_fused_and_staged_matching_agree(O, L, kps, model):
    """lsa_set_fused_match: one launch per ICP iteration (adaptive block search by the cell counts, per-lane sorted
    lists, model fit in the same kernel, tail kernel for what no block settles) against the staged kernels (first
    stage -> second stage -> model fit): histograms, status, weights and records bit for bit, for every keypoint type,
    ego-motion and localization parameters, several cell sizes and neighbour counts"""
    prev, cur = kps[model]
    a, b, c = L.Context(0), L.Context(0), L.Context(0)
    a.set_fused_match(1)  # one launch: every workgroup fits the models of the keypoints it has searched
    b.set_fused_match(0)
    c.set_fused_match(2)  # two launches: searches, then model fits
    cases = [
        (L.MatchParams.ego_motion(saturation_distance=5.0), (0.5, 0.25, 0.5)),
        (L.MatchParams.localization(saturation_distance=2.0), (0.75, 0.6, 0.3)),
        (L.MatchParams.localization(saturation_distance=2.0), (4.0, 3.0, 5.0)),
        (L.MatchParams.localization(saturation_distance=2.0), (0.1, 0.1, 0.1)),
    ]
    odd = L.MatchParams.localization(saturation_distance=1.0)
    odd.edge_nb_neighbors, odd.plane_nb_neighbors, odd.blob_nb_neighbors = 16, 7, 13
    cases.append((odd, (0.75, 0.6, 0.3)))
    for mp, cells in cases:
        for ctx in (a, b, c):
            for k in (L.EDGE, L.PLANE, L.BLOB):
                n = 20000 if k == L.BLOB else None
                ctx.set_keypoints(L.SET_WORKING, k, cur[k][:n])
                ctx.set_target(k, prev[k], cell=cells[k])
        ha = a.match_types(7, L.SET_WORKING, mp, perturbed())
        for other in (b, c):
            hb = other.match_types(7, L.SET_WORKING, mp, perturbed())
            assert ha.tolist() == hb.tolist()
            for k in (L.EDGE, L.PLANE, L.BLOB):
                sa, wa, ra = a.match_results(k, L.SET_WORKING)
                sb, wb, rb = other.match_results(k, L.SET_WORKING)
                assert np.array_equal(sa, sb), (k, cells, np.flatnonzero(sa != sb)[:10])
                assert np.array_equal(bits(wa), bits(wb)) and np.array_equal(bits(ra), bits(rb))
    # keypoints far from a sparse target: no block of the grid holds k points -> tail kernel (edges), FAR (planes)
    far = cur[L.EDGE][:500].copy()
    far["x"] += 300.0
    for ctx in (a, b, c):
        ctx.set_keypoints(L.SET_WORKING, L.EDGE, np.concatenate([cur[L.EDGE][:500], far]))
        ctx.set_target(L.EDGE, prev[L.EDGE][::7], cell=0.3)
        ctx.set_keypoints(L.SET_WORKING, L.PLANE, np.concatenate([cur[L.PLANE][:500], far]))
        ctx.set_target(L.PLANE, prev[L.PLANE][::50], cell=0.3)
    mp = L.MatchParams.localization(saturation_distance=2.0)
    ha = a.match_types(3, L.SET_WORKING, mp, np.eye(4))
    for other in (b, c):
        hb = other.match_types(3, L.SET_WORKING, mp, np.eye(4))
        assert ha.tolist() == hb.tolist()
        for k in (L.EDGE, L.PLANE):
            sa, wa, ra = a.match_results(k, L.SET_WORKING)
            sb, wb, rb = other.match_results(k, L.SET_WORKING)
            assert np.array_equal(sa, sb) and np.array_equal(bits(ra), bits(rb))
    # invalid parameters for one type: its keypoints are not searched, their status comes from the model kernel (the
    # one-launch form steps aside for that match)
    bad = L.MatchParams.localization(saturation_distance=2.0)
    bad.plane_nb_neighbors = 2
    ha = a.match_types(3, L.SET_WORKING, bad, np.eye(4))
    hb = b.match_types(3, L.SET_WORKING, bad, np.eye(4))
    assert ha.tolist() == hb.tolist() and ha[L.PLANE][1] > 0  # LSA_MATCH_BAD_MODEL_PARAMETRIZATION
    for ctx in (a, b, c):
        ctx.close()


def test_an_iteration_enqueued_behind_a_gate_equals_the_one_enqueued_in_line(gpu_ctx, O, L, kps):
    """lsa_icp_gate: a match and a solve enqueued BEFORE their pose exists wait on the device; posted, they compute what the
    same calls compute when they are enqueued with the pose; called off, they leave no trace."""
    prev, cur = kps[16]
    mp = L.MatchParams.ego_motion(saturation_distance=5.0)
    mp2 = L.MatchParams.ego_motion(saturation_distance=3.0)
    for k in (0, 1):
        gpu_ctx.set_keypoints(L.SET_WORKING, k, cur[k])
        gpu_ctx.set_target(k, prev[k])
    gpu_ctx.set_keypoints(L.SET_WORKING, 2, cur[2][:0])
    pose, w0 = perturbed(0.45, 0.01), np.array([0.45, 0.02, 0.0, 0.0, 0.0, 0.01])
    # in line
    gpu_ctx.match_types(3, L.SET_WORKING, mp, pose, histograms=False)
    want = gpu_ctx.solve_device(3, w0)
    want_match = [gpu_ctx.match_results(k, L.SET_WORKING) for k in (0, 1)]
    serial = [gpu_ctx.match_serial(k) for k in (0, 1)]
    # behind a gate, released after a while
    t = gpu_ctx.icp_gate()
    assert gpu_ctx.match_types_gated(3, L.SET_WORKING, mp) == 0
    gpu_ctx.solve_device_begin(3, None)
    time.sleep(0.005)
    gpu_ctx.icp_post(t, pose, w0)
    got = gpu_ctx.solve_device_end()
    assert list(got.pose) == list(want.pose) and got.final_cost == want.final_cost and got.num_evaluations == want.num_evaluations
    assert list(got.H) == list(want.H) and got.num_matches == want.num_matches
    for k in (0, 1):
        for a, b in zip(want_match[k], gpu_ctx.match_results(k, L.SET_WORKING)):
            assert a.tobytes() == b.tobytes()
    # called off: the launches do nothing, what they had announced is taken back
    serial = [gpu_ctx.match_serial(k) for k in (0, 1)]
    t = gpu_ctx.icp_gate()
    assert gpu_ctx.match_types_gated(3, L.SET_WORKING, mp2) == 0
    gpu_ctx.solve_device_begin(3, None)
    gpu_ctx.icp_cancel(t)
    gpu_ctx.solve_device_drop()
    gpu_ctx.sync()
    assert [gpu_ctx.match_serial(k) for k in (0, 1)] == serial
    for k in (0, 1):
        for a, b in zip(want_match[k], gpu_ctx.match_results(k, L.SET_WORKING)):
            assert a.tobytes() == b.tobytes()
    again = gpu_ctx.solve_device(3, w0)  # (the saturation distance of the match that did run is still in force)
    assert list(again.pose) == list(want.pose) and again.final_cost == want.final_cost
    # two iterations in the queue at once, the second one behind its gate while the first runs
    gpu_ctx.match_types(3, L.SET_WORKING, mp, pose, histograms=False)
    gpu_ctx.solve_device_begin(3, w0)
    t = gpu_ctx.icp_gate()
    assert gpu_ctx.match_types_gated(3, L.SET_WORKING, mp2) == 0
    gpu_ctx.solve_device_begin(3, None)
    first = gpu_ctx.solve_device_end()
    assert list(first.pose) == list(want.pose)
    pose2 = se3(*first.pose)
    gpu_ctx.icp_post(t, pose2, np.array(first.pose))
    second = gpu_ctx.solve_device_end()
    gpu_ctx.match_types(3, L.SET_WORKING, mp2, pose2, histograms=False)
    ref2 = gpu_ctx.solve_device(3, np.array(first.pose))
    assert list(second.pose) == list(ref2.pose) and second.final_cost == ref2.final_cost
    assert gpu_ctx.solve_device_fallbacks() == 0


def test_links_left_by_the_device_equal_the_host_algebra(gpu_ctx, O, L, kps):
    """lsa_icp_link: the solve itself leaves, on the device, what the iteration enqueued behind it reads -- run or not, the
    pose from its parameters, the next start point, the refined undistortion (Slam.cxx:940-950, 1134-1151, 1322-1352).
    (a) The block equals what lsa_posemath.h gives on the HOST from the same result, word for word (one source, two
    compilers), for ego-motion and localization links, with and without a trajectory to interpolate, when the solve made no
    step (go = 0) and when it was skipped.  (b) A whole loop enqueued at once -- three iterations behind links, the host only
    reading results -- computes what the same iterations compute one by one with the host in between."""
    prev, cur = kps[16]
    mp = L.MatchParams.ego_motion(saturation_distance=5.0)
    mp2 = L.MatchParams.ego_motion(saturation_distance=3.0)
    mp3 = L.MatchParams.ego_motion(saturation_distance=1.5)
    for k in (0, 1):
        gpu_ctx.set_keypoints(L.SET_WORKING, k, cur[k])
        gpu_ctx.set_target(k, prev[k])
    gpu_ctx.set_keypoints(L.SET_WORKING, 2, cur[2][:0])
    pose, w0 = perturbed(0.45, 0.01), np.array([0.45, 0.02, 0.0, 0.0, 0.0, 0.01])
    rng = np.random.default_rng(5)

    def link_for(refine, have_log, first=1, ratio=3.0):
        ln = L.IcpLink()
        ln.refine_undistortion, ln.first, ln.have_log = refine, first, have_log
        ln.prev_time, ln.cur_time, ln.max_extrapolation_ratio = 10.0, 10.1, ratio
        pw = se3(*(np.array([-0.5, 0.03, 0.01, 0.002, -0.004, 0.02]) + 1e-3 * rng.standard_normal(6)))
        ln.previous_world[:] = list(pw.reshape(-1))
        # a motion within the frame as an earlier refinement would have left it: two unit quaternions, two translations
        q0, q1 = rng.standard_normal(4) * 1e-3 + [1, 0, 0, 0], rng.standard_normal(4) * 1e-3 + [1, 0, 0, 0]
        q0, q1 = q0 / np.linalg.norm(q0), q1 / np.linalg.norm(q1)
        ln.motion[:] = [-0.1, 0.0] + list(q0) + list(q1) + list(1e-2 * rng.standard_normal(3)) + list(1e-2 * rng.standard_normal(3))
        return ln

    # (a) one solve, one link, the block read back
    cases = [(0, 0, 3.0), (1, 1, 3.0), (1, 0, 3.0), (1, 1, 0.5)]  # (the last: the frame's span is beyond MaxExtrapolationRatio)
    for refine, have_log, ratio in cases:
        ln = link_for(refine, have_log, ratio=ratio)
        gpu_ctx.match_types(3, L.SET_WORKING, mp, pose, histograms=False)
        t = gpu_ctx.icp_link()
        gpu_ctx.solve_device_begin_linked(3, w0, t, ln)
        res = gpu_ctx.solve_device_end()
        got = gpu_ctx.icp_link_peek(t)
        want, motion = L.icp_link_expected(res.pose, res.skipped, res.num_successful_steps, ln)
        assert want[0] == 1 and res.num_successful_steps > 1
        nwords = 1 + 12 + 6 + (0 if not refine else 64 - 19)
        assert np.array_equal(got[:nwords], want[:nwords]), (refine, have_log, ratio, np.nonzero(got[:nwords] != want[:nwords])[0])
        gpu_ctx.icp_abandon()
    # a solve that makes no step (no LM iteration allowed) and one that is skipped leave go = 0
    for max_iter, min_matches in ((0, 0), (15, 10**6)):
        gpu_ctx.match_types(3, L.SET_WORKING, mp, pose, histograms=False)
        ln = link_for(1, 1)
        t = gpu_ctx.icp_link()
        gpu_ctx.solve_device_begin_linked(3, w0, t, ln, max_iter=max_iter, min_matches=min_matches)
        res = gpu_ctx.solve_device_end()
        assert res.skipped == (1 if min_matches else 0) and (min_matches or res.num_successful_steps == 1)
        assert gpu_ctx.icp_link_peek(t)[0] == 0 == L.icp_link_expected(res.pose, res.skipped, res.num_successful_steps, ln)[0][0]
        gpu_ctx.icp_abandon()

    # (b) three iterations enqueued at once
    gpu_ctx.match_types(3, L.SET_WORKING, mp, pose, histograms=False)
    ln = link_for(0, 0)
    t1 = gpu_ctx.icp_link()
    gpu_ctx.solve_device_begin_linked(3, w0, t1, ln)
    assert gpu_ctx.match_types_gated(3, L.SET_WORKING, mp2) == 0
    t2 = gpu_ctx.icp_link()
    gpu_ctx.solve_device_begin_linked(3, None, t2, ln)
    assert gpu_ctx.match_types_gated(3, L.SET_WORKING, mp3) == 0
    gpu_ctx.solve_device_begin_linked(3, None, -1, None)
    chain = [gpu_ctx.solve_device_end() for _ in range(3)]
    chain_match = [gpu_ctx.match_results(k, L.SET_WORKING) for k in (0, 1)]
    gpu_ctx.icp_abandon()
    p, w, one_by_one = pose, w0, []
    for m in (mp, mp2, mp3):
        gpu_ctx.match_types(3, L.SET_WORKING, m, p, histograms=False)
        r = gpu_ctx.solve_device(3, w)
        one_by_one.append(r)
        words, _ = L.icp_link_expected(r.pose, r.skipped, r.num_successful_steps, ln)
        assert words[0] == 1
        d = words[1:19].view(np.float64)
        p, w = np.eye(4), d[12:18].copy()
        p[:3, :3], p[:3, 3] = d[:9].reshape(3, 3), d[9:12]
    for a, b in zip(chain, one_by_one):
        assert list(a.pose) == list(b.pose) and a.final_cost == b.final_cost and a.num_evaluations == b.num_evaluations and a.num_matches == b.num_matches
    for k in (0, 1):
        for a, b in zip(chain_match[k], gpu_ctx.match_results(k, L.SET_WORKING)):
            assert a.tobytes() == b.tobytes()
    assert gpu_ctx.solve_device_fallbacks() == 0


def test_a_solve_abandoned_on_the_device_says_so(gpu_ctx, O, L, kps):
    """lsa_debug_set("lm_give_up_block"): the workgroup stops exchanging sums, the others run into their 20 ms limit, the
    launch drains and lsa_solve_device reports LSA_E_STATE (never a wrong pose); the next solve is healthy again."""
    rec, st = setup_residuals(gpu_ctx, O, L, kps, 128)
    w0 = np.array([0.45, 0.02, 0.0, 0.0, 0.0, 0.01])
    want = gpu_ctx.solve_device(7, w0)
    before = gpu_ctx.solve_device_fallbacks()
    gpu_ctx.debug_set("lm_give_up_block", 3)
    with pytest.raises(L.LsaError, match="-4"):
        gpu_ctx.solve_device(7, w0)
    assert gpu_ctx.solve_device_fallbacks() == before + 1
    again = gpu_ctx.solve_device(7, w0)
    assert list(again.pose) == list(want.pose) and again.final_cost == want.final_cost
    # the host-driven loop the caller falls back to takes the same decisions (test_one_launch_solve_equals_the_host_driven_loop)
    pose, summ, costs = gpu_ctx.solve(7, perturbed(0.45, 0.01))
    assert (want.num_successful_steps, want.num_iterations) == (summ[0], summ[2])
