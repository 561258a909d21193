"""LidarSlam::RollingGrid on the device (lsa_device_grid_*: sorted voxel array, sort + fold + merge per Add,
stable compactions for Roll / ClearOldPoints / sub-maps) against the oracle's restatement of slam_lib/src/RollingGrid.cxx,
call by call, byte for byte -- map content, counts, sub-maps and their point order (key order on both sides).  The same
scenarios as tests/test_rolling_grid.py runs on the host grid."""
import numpy as np
import pytest

import lidarslam_amd as L
from oracle import oracle as O
from test_rolling_grid import FLT_MAX, cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def pair(ctx, **params):
    return L.DeviceGrid(ctx, **params), O.RollingGrid(**params)


def same_state(g, o):
    assert g.size() == o.size()
    a, b = g.get(), o.get()
    assert a.tobytes() == b.tobytes(), (a.size, b.size)
    assert g.get(clean=True).tobytes() == o.get(clean=True).tobytes()


def same_submap(g, o, mn=None, mx=None, min_nb=-1):
    na, nb = g.build_submap(mn, mx, min_nb), o.build_submap(mn, mx, min_nb)
    assert na == nb
    assert g.submap_valid() == o.submap_valid()
    assert g.ctx.target(L.PLANE).tobytes() == o.submap().tobytes()
    return na


@pytest.mark.parametrize("sampling", [0, 1, 2, 3, 4])  # FIRST, LAST, MAX_INTENSITY, CENTER_POINT, CENTROID
@pytest.mark.parametrize("min_frames", [0, 3])
def test_add_roll_and_submaps_follow_the_oracle(ctx, sampling, min_frames):
    rng = np.random.default_rng(100 + sampling * 7 + min_frames)
    g, o = pair(ctx, GridSize=12, VoxelResolution=8.0, LeafSize=0.6, Sampling=sampling, MinFramesPerVoxel=min_frames)
    total = 0
    for step in range(14):
        centre = np.array([step * 5.0, step * -2.0, 0.5 * step])
        pts = cloud(rng, 1500, centre, spread=14.0, t=step * 0.1, labels=True)
        if step % 4 == 1:
            pts = pts[:0]  # an empty keypoint cloud leaves the map alone (RollingGrid.cxx:119-123)
        for m in (g, o):
            m.add(pts, fixed=(step == 3), time=step * 0.1, roll=(step % 5 != 2))
        same_state(g, o)
        q = cloud(rng, 200, centre, spread=9.0)
        mn = np.array([q["x"].min(), q["y"].min(), q["z"].min()], np.float32)
        mx = np.array([q["x"].max(), q["y"].max(), q["z"].max()], np.float32)
        total += same_submap(g, o, mn, mx, min_nb=100)
        if step % 3 == 0:
            total += same_submap(g, o, mn, mx, min_nb=-1)
            total += same_submap(g, o)
    assert total > 1000
    g.close()


def test_large_batches_and_many_keyframes(ctx):
    """the sizes of the pipeline: tens of thousands of keypoints per keyframe, a map of a few hundred thousand voxels,
    duplicates inside a batch (several points per leaf voxel) and across batches"""
    rng = np.random.default_rng(11)
    g, o = pair(ctx, GridSize=50, VoxelResolution=10.0, LeafSize=0.6, Sampling=2)
    for step in range(12):
        centre = np.array([step * 3.0, 0.0, 0.0])
        pts = cloud(rng, 30000, centre, spread=40.0, t=step * 0.1)
        for m in (g, o):
            m.add(pts, time=step * 0.1)
        assert g.size() == o.size()
    same_state(g, o)
    q = cloud(rng, 500, centre, spread=30.0)
    mn = np.array([q["x"].min(), q["y"].min(), q["z"].min()], np.float32)
    mx = np.array([q["x"].max(), q["y"].max(), q["z"].max()], np.float32)
    assert same_submap(g, o, mn, mx, min_nb=250) > 50000
    g.close()


def test_the_box_of_an_empty_cloud_selects_nothing(ctx):
    rng = np.random.default_rng(3)
    g, o = pair(ctx, GridSize=20, VoxelResolution=5.0, LeafSize=0.4)
    pts = cloud(rng, 3000, np.zeros(3), spread=10.0)
    g.add(pts), o.add(pts)
    assert same_submap(g, o) == g.size() > 0
    assert g.submap_valid()
    mn, mx = np.full(3, FLT_MAX, np.float32), np.full(3, -FLT_MAX, np.float32)
    assert same_submap(g, o, mn, mx, min_nb=0) == 0
    assert not g.submap_valid()
    assert same_submap(g, o, np.full(3, 1e4, np.float32), np.full(3, 2e4, np.float32), 10) == 0
    assert same_submap(g, o, np.full(3, -1e8, np.float32), np.full(3, 1e8, np.float32), 10) == g.size()
    g.close()


def test_rolling_away_drops_the_voxels_left_behind(ctx):
    rng = np.random.default_rng(4)
    g, o = pair(ctx, GridSize=6, VoxelResolution=4.0, LeafSize=0.3)
    a = cloud(rng, 2000, np.zeros(3), spread=4.0)
    g.add(a), o.add(a)
    n0 = g.size()
    for shift in (3.0, 9.0, 40.0):
        mn, mx = np.full(3, shift - 1, np.float32), np.full(3, shift + 1, np.float32)
        g.roll(mn, mx), o.roll(mn, mx)
        same_state(g, o)
    assert g.size() == 0 < n0
    far = cloud(rng, 500, np.full(3, 500.0), spread=2.0)
    g.add(far, roll=False), o.add(far, roll=False)
    same_state(g, o)
    assert g.size() == 0
    g.add(far), o.add(far)
    same_state(g, o)
    tall = far.copy()
    tall["z"] -= np.linspace(0, 12, tall.size, dtype=np.float32)
    g.add(tall), o.add(tall)
    same_state(g, o)
    assert g.size() > 0
    g.close()


def test_decaying_threshold_and_fixed_points(ctx):
    rng = np.random.default_rng(5)
    g, o = pair(ctx, GridSize=10, VoxelResolution=6.0, LeafSize=0.5, DecayingThreshold=0.35, Sampling=1)
    for step in range(8):
        pts = cloud(rng, 800, np.array([step * 1.0, 0, 0]), spread=8.0, t=step * 0.1)
        for m in (g, o):
            m.add(pts, fixed=(step == 1), time=step * 0.1)
        same_state(g, o)
    before = g.size()
    for m in (g, o):
        m.clear_old_points(1.0)
    same_state(g, o)
    assert 0 < g.get().size < before == g.size()
    same_submap(g, o)
    # Size() after the decay: an Add that does not move the grid keeps counting from the old figure (Roll returns before
    # its recount, RollingGrid.cxx:136-137), one that moves it recounts
    pts = cloud(rng, 800, np.array([8.0, 0, 0]), spread=8.0, t=1.0)
    for m in (g, o):
        m.add(pts, time=1.0)
    same_state(g, o)
    pts = cloud(rng, 800, np.array([60.0, 0, 0]), spread=8.0, t=1.1)
    for m in (g, o):
        m.add(pts, time=1.1)
    same_state(g, o)
    g.close()


@pytest.mark.parametrize("n", [1, 255, 4096, 4097, 12289])
def test_batches_around_the_sizes_the_sort_is_built_of(ctx, n):
    """one point, just under a block, exactly one run of the LDS sort, one more, three runs and a bit"""
    rng = np.random.default_rng(n)
    g, o = pair(ctx, GridSize=30, VoxelResolution=6.0, LeafSize=0.5, Sampling=1)
    for step in range(3):
        pts = cloud(rng, n, np.array([step * 7.0, 0, 0]), spread=25.0, t=step * 0.1)
        for m in (g, o):
            m.add(pts, time=step * 0.1)
        same_state(g, o)
    g.close()


def test_changing_the_geometry_puts_the_points_back(ctx):
    rng = np.random.default_rng(6)
    g, o = pair(ctx, GridSize=10, VoxelResolution=6.0, LeafSize=0.5)
    pts = cloud(rng, 1000, np.zeros(3), spread=8.0)
    for name, value in (("GridSize", 14), ("VoxelResolution", 3.0), ("LeafSize", 0.25)):
        g.add(pts), o.add(pts)
        assert g.size() == o.size() > 0
        g.set(name, value), o.set(name, value)
        same_state(g, o)
    g.add(pts), o.add(pts)
    same_state(g, o)
    g.reset([1.0, 2.0, 3.0]), o.reset([1.0, 2.0, 3.0])
    same_state(g, o)
    assert g.size() == 0
    g.add(pts, roll=False), o.add(pts, roll=False)
    same_state(g, o)
    g.clear(), o.clear()
    assert g.size() == o.size() == 0
    with pytest.raises(L.LsaError):
        g.set("NoSuchParameter", 1.0)
    g.close()


def test_centroid_sampling_with_the_reference_loop_quirk(ctx):
    """CENTROID (RollingGrid.cxx:263-297): the block that pulls every touched voxel towards the mean of its new points sits
    INSIDE the loop over the cloud, so a voxel's point depends on how many points of the cloud follow its own -- in arrival
    order, whatever voxel they fall into.  Dense batches (many points per leaf voxel, thousands of points behind a voxel's
    last one), fixed keyframes, points outside the grid, a batch that only revisits old voxels: byte for byte."""
    rng = np.random.default_rng(21)
    g, o = pair(ctx, GridSize=10, VoxelResolution=6.0, LeafSize=0.8, Sampling=4)
    centre = np.zeros(3)
    for step in range(10):
        n = (300, 5000, 20000, 257, 4097, 1)[step % 6]
        pts = cloud(rng, n, centre, spread=6.0 if step % 3 else 40.0, t=step * 0.1, labels=True)  # spread 40: most points outside the grid
        for m in (g, o):
            m.add(pts, fixed=(step == 4), time=step * 0.1, roll=(step in (0, 7)))
        same_state(g, o)
    # the same cloud again: no new voxel, every point pulls
    for m in (g, o):
        m.add(pts, time=1.5, roll=False)
    same_state(g, o)
    assert g.size() > 500
    g.close()


def test_keypoints_of_the_context_go_into_the_map_without_leaving_the_device(ctx):
    """lsa_device_grid_add_keypoints (Slam::UpdateMapsUsingTworld): the WORKING keypoints moved by the pose are what the
    oracle's grid gets from the host-side transform"""
    pts, _ = L.synth_frame(16, 1000, 0)
    ctx.upload_frame(pts)
    ctx.extract_keypoints()
    ctx.reset_working_keypoints()
    T = np.eye(4)
    T[:3, 3] = [3.0, -1.0, 0.2]
    c, s = np.cos(0.3), np.sin(0.3)
    T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    g, o = pair(ctx, LeafSize=0.6)
    g.add_keypoints(L.SET_WORKING, L.PLANE, T, 0.5)
    o.add(O.transform(ctx.keypoints(L.SET_WORKING, L.PLANE), T), time=0.5)
    same_state(g, o)
    assert g.size() > 500
    g.close()


def test_the_two_step_calls_of_the_pipeline(ctx):
    """what lsa_slam does per keyframe: stage the keypoints on the caller's thread, insert them from ANOTHER thread on the
    grid's own stream, then sub-maps of several grids side by side whose boxes are the keypoints' boxes left on the device
    (nothing read back) -- same maps and sub-maps as the one-call forms and as the oracle"""
    import threading

    pts, _ = L.synth_frame(16, 1000, 0)
    ctx.upload_frame(pts)
    ctx.extract_keypoints()
    ctx.reset_working_keypoints()
    grids = {k: pair(ctx, LeafSize=leaf) for k, leaf in ((L.EDGE, 0.3), (L.PLANE, 0.6))}
    for step in range(3):
        T = np.eye(4)
        T[:3, 3] = [2.0 * step, 0.5 * step, 0.0]
        for k, (g, o) in grids.items():
            g.stage_keypoints(L.SET_WORKING, k, T)
        workers = [threading.Thread(target=g.add_staged, args=(0.1 * step,)) for g, _ in grids.values()]
        for w in workers:
            w.start()
        for w in workers:
            w.join()
        for k, (g, o) in grids.items():
            o.add(O.transform(ctx.keypoints(L.SET_WORKING, k), T), time=0.1 * step)
            same_state(g, o)
    # the box of the keypoints under a pose a little off the last keyframe's
    T[:3, 3] += [25.0, 0.0, 0.0]
    mn, mx = ctx.keypoint_bboxes(L.SET_WORKING, T)
    ctx._check(ctx.L.lsa_keypoint_bboxes_begin(ctx.h, L.SET_WORKING, L.ptr(L.pose16(T))), "lsa_keypoint_bboxes_begin")
    for k, (g, o) in grids.items():
        g.build_submap_begin_for_keypoints(k, 50, ktype=k)
    for k, (g, o) in grids.items():
        n = g.build_submap_end()
        assert n == o.build_submap(mn[k], mx[k], 50) and 0 < n < g.size()
        assert ctx.target(k).tobytes() == o.submap().tobytes()
        assert g.submap_valid()
        g.close()
