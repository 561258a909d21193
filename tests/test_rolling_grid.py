"""the host RollingGrid behind the C ABI against the oracle's restatement of
slam_lib/src/RollingGrid.cxx, call by call.  The order of the points matters (it fixes the
summation order of every PCA fed from the sub-map), so clouds are compared byte for byte.
No GPU involved: these run in the CPU suite.  Parity unpinned (the reference holds no fixture
for RollingGrid; its own test is the end-to-end trajectory comparison)."""
import numpy as np
import pytest

import lidarslam_amd as L
from oracle import oracle as O

FLT_MAX = np.finfo(np.float32).max


def cloud(rng, n, centre, spread=30.0, t=0.0, labels=False):
    p = np.zeros(n, L.POINT_DTYPE)
    xyz = (rng.normal(size=(n, 3)) * [spread, spread, spread / 6] + centre).astype(np.float32)
    p["x"], p["y"], p["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    p["intensity"] = rng.integers(0, 255, n).astype(np.float32)
    p["time"] = t + rng.random(n) * 0.1
    p["laser_id"] = rng.integers(0, 64, n)
    if labels:
        p["label"] = (rng.random(n) < 0.1).astype(p["label"].dtype)
    return p


def pair(**params):
    return L.RollingGrid(**params), O.RollingGrid(**params)


def same_state(g, o):
    assert g.size() == o.size()
    a, b = g.get(), o.get()
    assert a.tobytes() == b.tobytes()
    a, b = g.get(clean=True), o.get(clean=True)
    assert a.tobytes() == b.tobytes()


def same_submap(g, o, mn=None, mx=None, min_nb=-1):
    na, nb = g.build_submap(mn, mx, min_nb), o.build_submap(mn, mx, min_nb)
    assert na == nb
    assert g.submap_valid() == o.submap_valid()
    assert g.submap().tobytes() == o.submap().tobytes()
    return na


@pytest.mark.parametrize("sampling", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("min_frames", [0, 3])
def test_add_roll_and_submaps_follow_the_oracle(sampling, min_frames):
    rng = np.random.default_rng(100 + sampling * 7 + min_frames)
    g, o = pair(GridSize=12, VoxelResolution=8.0, LeafSize=0.6, Sampling=sampling, MinFramesPerVoxel=min_frames)
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


def test_the_box_of_an_empty_cloud_selects_nothing():
    """pcl::getMinMax3D of an empty cloud is (+FLT_MAX, -FLT_MAX): the sub-map is empty and counts as invalid,
    so that the next frame rebuilds it (Slam.cxx:1013; RollingGrid.h:154)."""
    rng = np.random.default_rng(3)
    g, o = pair(GridSize=20, VoxelResolution=5.0, LeafSize=0.4)
    pts = cloud(rng, 3000, np.zeros(3), spread=10.0)
    g.add(pts), o.add(pts)
    assert same_submap(g, o) == g.size() > 0
    assert g.submap_valid()
    mn, mx = np.full(3, FLT_MAX, np.float32), np.full(3, -FLT_MAX, np.float32)
    assert same_submap(g, o, mn, mx, min_nb=0) == 0
    assert not g.submap_valid()
    # a box outside of the grid
    assert same_submap(g, o, np.full(3, 1e4, np.float32), np.full(3, 2e4, np.float32), 10) == 0
    # a huge box is clipped to the grid: everything (beyond the int range the reference's cast is undefined)
    assert same_submap(g, o, np.full(3, -1e8, np.float32), np.full(3, 1e8, np.float32), 10) == g.size()


def test_rolling_away_drops_the_voxels_left_behind():
    rng = np.random.default_rng(4)
    g, o = pair(GridSize=6, VoxelResolution=4.0, LeafSize=0.3)
    a = cloud(rng, 2000, np.zeros(3), spread=4.0)
    g.add(a), o.add(a)
    n0 = g.size()
    for shift in (3.0, 9.0, 40.0):
        mn, mx = np.full(3, shift - 1, np.float32), np.full(3, shift + 1, np.float32)
        g.roll(mn, mx), o.roll(mn, mx)
        same_state(g, o)
    assert g.size() == 0 < n0
    # points outside of the grid are ignored when the grid must not move
    far = cloud(rng, 500, np.full(3, 500.0), spread=2.0)
    g.add(far, roll=False), o.add(far, roll=False)
    same_state(g, o)
    assert g.size() == 0
    # the grid follows the cloud (Roll stops as soon as the upper faces meet, and voxel k covers
    # origin + (k -+ 0.5) * resolution: a thin cloud can still end up above the last voxel layer)
    g.add(far), o.add(far)
    same_state(g, o)
    tall = far.copy()
    tall["z"] -= np.linspace(0, 12, tall.size, dtype=np.float32)
    g.add(tall), o.add(tall)
    same_state(g, o)
    assert g.size() > 0


def test_decaying_threshold_and_fixed_points():
    rng = np.random.default_rng(5)
    g, o = pair(GridSize=10, VoxelResolution=6.0, LeafSize=0.5, DecayingThreshold=0.35, Sampling=1)
    for step in range(8):
        pts = cloud(rng, 800, np.array([step * 1.0, 0, 0]), spread=8.0, t=step * 0.1)
        for m in (g, o):
            m.add(pts, fixed=(step == 1), time=step * 0.1)
        same_state(g, o)
    before = g.size()
    for m in (g, o):
        m.clear_old_points(1.0)
    same_state(g, o)
    # the fixed points of step 1 stay, old moving ones are gone; Size() is not told (RollingGrid.cxx:325-351)
    assert 0 < g.get().size < before == g.size()
    same_submap(g, o)


def test_changing_the_geometry_clears_the_map():
    rng = np.random.default_rng(6)
    g, o = pair(GridSize=10, VoxelResolution=6.0, LeafSize=0.5)
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


@pytest.mark.parametrize("sampling", [0, 1, 2, 3, 4])
def test_add_on_several_threads_builds_the_same_map(sampling):
    """"AddThreads" is an implementation knob: outer voxels are created in point order by one thread, the leaf voxels of
    each outer voxel are filled by one thread in point order -- same content, same iteration order, same sub-maps as the
    oracle's sequential loop (CENTROID couples the voxels and stays sequential)."""
    rng = np.random.default_rng(40 + sampling)
    g, o = pair(GridSize=16, VoxelResolution=6.0, LeafSize=0.5, Sampling=sampling, MinFramesPerVoxel=2)
    g.set("AddThreads", 4)
    for step in range(10):
        centre = np.array([step * 4.0, step * 1.5, 0.0])
        pts = cloud(rng, 6000 if step != 4 else 500, centre, spread=16.0, t=step * 0.1, labels=True)
        for m in (g, o):
            m.add(pts, fixed=(step == 2), time=step * 0.1)
        same_state(g, o)
        q = cloud(rng, 300, centre, spread=12.0)
        mn = np.array([q["x"].min(), q["y"].min(), q["z"].min()], np.float32)
        mx = np.array([q["x"].max(), q["y"].max(), q["z"].max()], np.float32)
        assert same_submap(g, o, mn, mx, min_nb=150) > 0
    # the sub-map extraction shares the helper threads once the map is big (all points of the voxels the box touches)
    assert g.size() >= 8192 or sampling == 4
    g.set("MinFramesPerVoxel", 0), o.set("MinFramesPerVoxel", 0)
    assert same_submap(g, o, mn, mx, min_nb=150) > 0
    assert same_submap(g, o, np.full(3, -1e6, np.float32), np.full(3, 1e6, np.float32), 10) == g.size()
    g.set("AddThreads", 1)
    pts = cloud(rng, 6000, np.zeros(3), spread=16.0)
    g.add(pts), o.add(pts)
    same_state(g, o)


def test_the_order_of_the_reference_containers_is_still_available():
    """"Ordered" = 0: Get / BuildSubMap hand the voxels out in the iteration order of the reference's own
    unordered_map<int, unordered_map<int, Voxel>> (libstdc++), host grid and oracle alike; the default is key order"""
    rng = np.random.default_rng(9)
    g, o = pair(GridSize=12, VoxelResolution=8.0, LeafSize=0.6)
    g.set("Ordered", 0), o.set("Ordered", 0)
    g2, o2 = pair(GridSize=12, VoxelResolution=8.0, LeafSize=0.6)
    for step in range(5):
        pts = cloud(rng, 3000, np.array([step * 4.0, 0, 0]), spread=12.0, t=step * 0.1)
        for m in (g, o, g2, o2):
            m.add(pts, time=step * 0.1)
    same_state(g, o)
    same_state(g2, o2)
    a, b = g.get(), g2.get()
    assert a.tobytes() != b.tobytes() and np.array_equal(np.sort(a, order=["x", "y", "z"]), np.sort(b, order=["x", "y", "z"]))
    mn, mx = np.array([0, -10, -5], np.float32), np.array([20, 10, 5], np.float32)
    assert same_submap(g, o, mn, mx) == same_submap(g2, o2, mn, mx) > 0
