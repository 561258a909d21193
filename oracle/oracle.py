"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see orc_math.hpp).

ctypes front-end of oracle/liboracle.so, the CPU restatement of the reference's
scan-matching hot path.  Importable only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; the product never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

import sys

sys.path.insert(0, os.path.dirname(HERE))
from lidarslam_amd._native import POINT_DTYPE, ExtractParams, MatchParams, ptr, pose16  # noqa: E402


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".cpp", ".hpp"))]
    srcs += [os.path.join(HERE, "..", "include", f) for f in ("lsa_pmath.h", "lidarslam_amd.h")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "liboracle.so"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
        L.orc_extractor_create.restype = vp
        L.orc_extractor_destroy.argtypes = [vp]
        L.orc_extractor_set_threads.argtypes = [vp, i32]
        L.orc_extractor_get_azimuthal_resolution.restype = C.c_float
        L.orc_extractor_get_azimuthal_resolution.argtypes = [vp]
        L.orc_extractor_set_azimuthal_resolution.argtypes = [vp, C.c_float]
        L.orc_extractor_compute.argtypes = [vp, C.POINTER(ExtractParams), vp, i32, vp]
        L.orc_extractor_keypoints.argtypes = [vp, i32, vp, i32]
        L.orc_extractor_debug.argtypes = [vp, i32, vp, i32]
        L.orc_extractor_nb_rings.argtypes = [vp]
        L.orc_knn.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp]
        L.orc_knn_brute.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp]
        L.orc_match.argtypes = [vp, i32, vp, i32, i32, C.POINTER(MatchParams), vp, i32, vp, vp, vp, vp]
        L.orc_accumulate.argtypes = [vp, vp, i32, f64, vp, i32, vp, vp, vp, vp]
        L.orc_lm_solve.argtypes = [vp, vp, i32, f64, vp, i32, i32, vp, vp, vp, vp]
        L.orc_covariance.argtypes = [vp, vp, i32, f64, vp, vp, vp]
        L.orc_math.argtypes = [i32, vp, vp, i32, vp]
        L.orc_undistort.argtypes = [vp, i32, vp, vp, f64, f64]
        L.orc_icp_link.argtypes = [vp, i32, i32, f64, f64, f64, vp, vp, vp, vp]
        L.orc_transform.argtypes = [vp, i32, vp]
        L.orc_slam_create.restype = vp
        L.orc_slam_destroy.argtypes = [vp]
        L.orc_slam_reset.argtypes = [vp, i32]
        L.orc_slam_set_param.argtypes = [vp, C.c_char_p, f64]
        L.orc_slam_add_frame.argtypes = [vp, vp, i32, C.c_uint64, C.c_uint32]
        L.orc_slam_get_world_transform.argtypes = [vp, vp, vp]
        L.orc_slam_get_covariance.argtypes = [vp, vp]
        L.orc_slam_get_keypoints.argtypes = [vp, i32, i32, vp, i32]
        L.orc_slam_get_registered_frame.argtypes = [vp, vp, i32]
        L.orc_slam_get_match_status.argtypes = [vp, i32, i32, vp, vp, i32]
        L.orc_slam_get_stats.argtypes = [vp, vp]
        L.orc_slam_get_submap.argtypes = [vp, i32, vp, i32]
        L.orc_slam_get_latency_compensated_world_transform.argtypes = [vp, vp]
        L.orc_slam_set_world_transform_from_guess.argtypes = [vp, vp]
        L.orc_slam_set_base_to_lidar_offset.argtypes = [vp, vp, i32]
        L.orc_slam_set_extractor_param.argtypes = [vp, i32, C.c_char_p, f64]
        L.orc_slam_add_frames.argtypes = [vp, vp, vp, vp, i32]
        L.orc_slam_get_trajectory.argtypes = [vp, vp, vp, i32]
        L.orc_slam_get_debug_information.argtypes = [vp, vp]
        L.orc_slam_get_map.argtypes = [vp, i32, i32, vp, i32]
        L.orc_grid_create.restype = vp
        L.orc_grid_destroy.argtypes = [vp]
        L.orc_grid_destroy.restype = None
        L.orc_grid_set.argtypes = [vp, C.c_char_p, f64]
        L.orc_grid_reset.argtypes = [vp, vp]
        L.orc_grid_reset.restype = None
        L.orc_grid_clear.argtypes = [vp]
        L.orc_grid_clear.restype = None
        L.orc_grid_size.argtypes = [vp]
        L.orc_grid_roll.argtypes = [vp, vp, vp]
        L.orc_grid_roll.restype = None
        L.orc_grid_add.argtypes = [vp, vp, i32, i32, f64, i32]
        L.orc_grid_add.restype = None
        L.orc_grid_clear_old_points.argtypes = [vp, f64]
        L.orc_grid_clear_old_points.restype = None
        L.orc_grid_get.argtypes = [vp, i32, vp, i32]
        L.orc_grid_build_submap.argtypes = [vp, vp, vp, i32]
        L.orc_grid_submap_valid.argtypes = [vp]
        L.orc_grid_submap.argtypes = [vp, vp, i32]
        _lib = L
    return _lib


DEBUG_INFORMATION_NAMES = [
    "EgoMotion: edges used", "EgoMotion: planes used", "Localization: edges used", "Localization: planes used",
    "Localization: blobs used", "Localization: position error", "Localization: orientation error",
    "Confidence: overlap", "Confidence: comply motion limits", "latency",
]

DEBUG_NAMES = [
    "sin_angle", "saliency", "depth_gap", "intensity_gap", "edge_keypoint", "plane_keypoint", "blob_keypoint",
    "edge_validity", "plane_validity", "blob_validity",
]


class Extractor:
    def __init__(self, threads=1):
        self.h = lib().orc_extractor_create()
        lib().orc_extractor_set_threads(self.h, threads)
        self.n = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_extractor_destroy(self.h)
            self.h = None

    @property
    def azimuthal_resolution(self):
        return lib().orc_extractor_get_azimuthal_resolution(self.h)

    @azimuthal_resolution.setter
    def azimuthal_resolution(self, v):
        lib().orc_extractor_set_azimuthal_resolution(self.h, v)

    def compute(self, pts, params=None):
        params = params or ExtractParams()
        pts = np.ascontiguousarray(pts)
        counts = np.zeros(3, np.int32)
        lib().orc_extractor_compute(self.h, C.byref(params), ptr(pts), pts.size, ptr(counts))
        self.n = pts.size
        self.counts = counts
        return counts

    def keypoints(self, k):
        out = np.zeros(int(self.counts[k]), POINT_DTYPE)
        lib().orc_extractor_keypoints(self.h, k, ptr(out), out.size)
        return out

    def debug(self, i):
        out = np.zeros(self.n, np.float32)
        lib().orc_extractor_debug(self.h, i, ptr(out), out.size)
        return out

    def nb_rings(self):
        return lib().orc_extractor_nb_rings(self.h)


def knn(target, queries, k, brute=False):
    target = np.ascontiguousarray(target)
    q = np.ascontiguousarray(queries, np.float64).reshape(-1, 3)
    idx = np.full((q.shape[0], k), -1, np.int32)
    d2 = np.zeros((q.shape[0], k), np.float32)
    cnt = np.zeros(q.shape[0], np.int32)
    f = lib().orc_knn_brute if brute else lib().orc_knn
    f(ptr(target), target.size, ptr(q), q.shape[0], k, ptr(idx), ptr(d2), ptr(cnt))
    return idx, d2, cnt


def match(cur, tgt, ktype, mp, pose, threads=1):
    cur = np.ascontiguousarray(cur)
    tgt = np.ascontiguousarray(tgt)
    n = cur.size
    status = np.zeros(n, np.uint8)
    weights = np.zeros(n, np.float64)
    records = np.zeros((n, 16), np.float64)
    hist = np.zeros(8, np.int32)
    lib().orc_match(ptr(cur), n, ptr(tgt), tgt.size, ktype, C.byref(mp), ptr(pose16(pose)), threads, ptr(status), ptr(weights),
                    ptr(records), ptr(hist))
    return status, weights, records, hist


def accumulate(records, status, sat, w, jac=True):
    records = np.ascontiguousarray(records, np.float64)
    status = np.ascontiguousarray(status, np.uint8)
    w = np.ascontiguousarray(w, np.float64)
    cost = C.c_double()
    nv = C.c_int()
    g = np.zeros(6)
    H = np.zeros((6, 6))
    lib().orc_accumulate(ptr(records), ptr(status), status.size, sat, ptr(w), int(jac), C.byref(cost), ptr(g), ptr(H), C.byref(nv))
    return cost.value, g, H, nv.value


def lm_solve(records, status, sat, pose, max_iter=15, two_d=False):
    records = np.ascontiguousarray(records, np.float64)
    status = np.ascontiguousarray(status, np.uint8)
    out = np.zeros(16)
    w = np.zeros(6)
    summ = np.zeros(4, np.int32)
    costs = np.zeros(2)
    lib().orc_lm_solve(ptr(records), ptr(status), status.size, sat, ptr(pose16(pose)), max_iter, int(two_d), ptr(out), ptr(w), ptr(summ),
                       ptr(costs))
    return out.reshape(4, 4), w, summ, costs


def covariance(records, status, sat, pose):
    records = np.ascontiguousarray(records, np.float64)
    status = np.ascontiguousarray(status, np.uint8)
    cov = np.zeros((6, 6))
    err = np.zeros(2)
    lib().orc_covariance(ptr(records), ptr(status), status.size, sat, ptr(pose16(pose)), ptr(cov), ptr(err))
    return cov, err


def set_libm_trig(on):
    """the oracle's eigen-solver / slerp trigonometry: libm (as the reference's PCL / Eigen) instead of lsa_pmath.h"""
    lib().orc_set_libm_trig(int(bool(on)))


def math(fn, x, y=None):
    """Same function ids as lsa_selftest_math, evaluated on the host."""
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(x if y is None else y, np.float64)
    out = np.zeros_like(x)
    lib().orc_math(fn, ptr(x), ptr(y), x.size, ptr(out))
    return out


def undistort(pts, H0, H1, t0, t1):
    out = np.ascontiguousarray(pts).copy()
    lib().orc_undistort(ptr(out), out.size, ptr(pose16(H0)), ptr(pose16(H1)), t0, t1)
    return out


def icp_link(x6, refine, have_log, prev_time, cur_time, max_ratio, previous_world, motion):
    """orc_icp_link: the pose algebra between two ICP iterations by the restatement's own functions, laid out as the
    product's link block -> (64 words, the motion within the frame afterwards)"""
    words, after = np.zeros(64, np.uint64), np.zeros(16, np.float64)
    x, m = np.ascontiguousarray(x6, np.float64), np.ascontiguousarray(motion, np.float64)
    lib().orc_icp_link(ptr(x), int(refine), int(have_log), prev_time, cur_time, max_ratio, ptr(pose16(previous_world)), ptr(m), ptr(words), ptr(after))
    return words, after


def transform(pts, T):
    out = np.ascontiguousarray(pts).copy()
    lib().orc_transform(ptr(out), out.size, ptr(pose16(T)))
    return out


def velodyne_to_lidar(records, layout, mapping=None, device_id=0, rpm=600.0, timestamp_first_packet=False):
    """VelodyneToLidarNode::Callback on driver records; returns (LidarPoint array, time field was usable)."""
    rec = np.ascontiguousarray(records)
    n = rec.nbytes // int(layout[0])
    out = np.zeros(n, POINT_DTYPE)
    lay = (C.c_int32 * 7)(*[int(v) for v in layout])
    mp = np.ascontiguousarray(mapping, np.uint16) if mapping is not None else None
    f = lib().orc_velodyne_to_lidar
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
    rc = f(rec.ctypes.data_as(C.c_void_p), n, lay, ptr(mp) if mp is not None else None, 0 if mp is None else mp.size, device_id, rpm,
           int(timestamp_first_packet), ptr(out))
    return out, rc == 1


def robosense_to_lidar(records, width, height, layout, mapping=None, device_id=0, rpm=600.0):
    """RobosenseToLidarNode::Callback on the driver's organized cloud (height lasers x width points, pcl::PointXYZI records);
    returns the LidarPoint array of the points kept."""
    rec = np.ascontiguousarray(records)
    out = np.zeros(max(width * height, 1), POINT_DTYPE)
    lay = (C.c_int32 * 5)(*[int(v) for v in layout])
    mp = np.ascontiguousarray(mapping, np.uint16) if mapping is not None else None
    f = lib().orc_robosense_to_lidar
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
    n = f(rec.ctypes.data_as(C.c_void_p), width, height, lay, ptr(mp) if mp is not None else None, 0 if mp is None else mp.size, device_id, rpm, ptr(out))
    return out[: max(n, 0)].copy()


def polydata_to_point_cloud(xyz, time, laser_id, intensity, mapping=None, time_to_seconds=1.0):
    """vtkSlam::PolyDataToPointCloud (paraview_wrapping/Plugin/vtkLidarSlam/vtkSlam.cxx:668-707), numpy restatement:
    frame end = max of the time array (:682), stamp = end * (factor * 1e6) as an integer (:683), points with all-zero
    coordinates dropped (:691), time = (t - end) * factor (:697), laser_id through the mapping (:698).
    Returns (LidarPoint array, stamp_us, allPointsAreValid)."""
    xyz = np.asarray(xyz, np.float64).reshape(-1, 3)  # poly->GetPoint(i, double pos[3])
    t = np.asarray(time, np.float64)                   # GetTuple1 returns double whatever the array holds
    end = t.max()
    stamp = int(np.uint64(end * (time_to_seconds * 1e6)))
    keep = (xyz != 0).any(axis=1)
    out = np.zeros(int(keep.sum()), POINT_DTYPE)
    out["x"], out["y"], out["z"], out["w"] = xyz[keep, 0], xyz[keep, 1], xyz[keep, 2], 1.0
    out["time"] = (t[keep] - end) * time_to_seconds
    lid = np.asarray(laser_id, np.float64)[keep]
    out["laser_id"] = (np.asarray(mapping)[lid.astype(np.int64)] if mapping is not None else lid).astype(np.uint16)
    out["intensity"] = np.asarray(intensity, np.float64)[keep]
    return out, stamp, bool(keep.all())


def lcp(cloud, ratio, targets, leaves):
    """Confidence::LCPEstimator on a registered cloud; targets / leaves: per keypoint type (None = map not used)."""
    cloud = np.ascontiguousarray(cloud)
    tg = [np.ascontiguousarray(t) if t is not None else None for t in targets]
    arr = (C.c_void_p * 3)(*[t.ctypes.data if t is not None and t.size else None for t in tg])
    m = (C.c_int * 3)(*[int(t.size) if t is not None else 0 for t in tg])
    lf = (C.c_double * 3)(*[float(x) for x in leaves])
    f = lib().orc_lcp
    f.restype = C.c_float
    f.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    return float(f(ptr(cloud), cloud.size, ratio, arr, m, lf))


class Slam:
    """Mirror of the product's lidarslam_amd.Slam for the oracle."""

    def __init__(self, **params):
        self.h = lib().orc_slam_create()
        for k, v in params.items():
            self.set_param(k, v)

    def overlap(self):
        f = lib().orc_slam_get_overlap
        f.restype = C.c_float
        f.argtypes = [C.c_void_p]
        return float(f(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_slam_destroy(self.h)
            self.h = None

    def set_param(self, name, value):
        if lib().orc_slam_set_param(self.h, name.encode(), float(value)) != 0:
            raise KeyError(name)

    def reset(self, reset_log=True):
        lib().orc_slam_reset(self.h, int(reset_log))

    def add_frame(self, pts, stamp_us, seq=0):
        pts = np.ascontiguousarray(pts)
        self._n = pts.size
        lib().orc_slam_add_frame(self.h, ptr(pts), pts.size, stamp_us, seq)

    def world_transform(self):
        T = np.zeros(16)
        t = C.c_double()
        lib().orc_slam_get_world_transform(self.h, ptr(T), C.byref(t))
        return T.reshape(4, 4)

    def covariance(self):
        c = np.zeros((6, 6))
        lib().orc_slam_get_covariance(self.h, ptr(c))
        return c

    def keypoints(self, k, which=0, cap=400000):
        """which: 0 undistorted BASE, 1 WORLD, 2 raw BASE."""
        out = np.zeros(cap, POINT_DTYPE)
        n = lib().orc_slam_get_keypoints(self.h, k, which, ptr(out), cap)
        return out[:n].copy()

    def registered_frame(self):
        out = np.zeros(self._n, POINT_DTYPE)
        n = lib().orc_slam_get_registered_frame(self.h, ptr(out), out.size)
        return out[:n]

    def match_status(self, localization, k, cap=400000):
        st = np.zeros(cap, np.uint8)
        w = np.zeros(cap)
        n = lib().orc_slam_get_match_status(self.h, int(localization), k, ptr(st), ptr(w), cap)
        return st[:n].copy(), w[:n].copy()

    def stats(self):
        o = np.zeros(16)
        lib().orc_slam_get_stats(self.h, ptr(o))
        return o

    def submap(self, k, cap=4000000):
        out = np.zeros(cap, POINT_DTYPE)
        n = lib().orc_slam_get_submap(self.h, k, ptr(out), cap)
        return out[:n].copy()

    def map(self, k, clean=False, cap=4000000):
        out = np.zeros(cap, POINT_DTYPE)
        n = lib().orc_slam_get_map(self.h, k, int(clean), ptr(out), cap)
        return out[:n].copy()

    def latency_compensated_world_transform(self):
        T = np.zeros(16)
        lib().orc_slam_get_latency_compensated_world_transform(self.h, ptr(T))
        return T.reshape(4, 4)

    def set_world_transform_from_guess(self, T):
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        lib().orc_slam_set_world_transform_from_guess(self.h, ptr(T))

    def set_base_to_lidar_offset(self, T, device_id=0):
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        lib().orc_slam_set_base_to_lidar_offset(self.h, ptr(T), device_id)

    def set_extractor_param(self, device_id, name, value):
        assert lib().orc_slam_set_extractor_param(self.h, device_id, name.encode(), float(value)) == 0, name

    def add_frames(self, frames, stamps_us, seq=0):
        frames = [np.ascontiguousarray(f, POINT_DTYPE) for f in frames]
        ptrs = (C.c_void_p * len(frames))(*[f.ctypes.data if f.size else None for f in frames])
        sizes = np.array([f.size for f in frames], np.int32)
        stamps = np.array(stamps_us, np.uint64)
        self._n = max(getattr(self, "_n", 0), int(sizes.sum()))
        lib().orc_slam_add_frames(self.h, ptrs, ptr(sizes), ptr(stamps), len(frames))

    def trajectory(self, cap=100000):
        """(n, 4, 4) poses, (n,) times, (n, 6, 6) covariances"""
        n = lib().orc_slam_get_trajectory(self.h, None, None, 0)
        rows, cov = np.zeros((max(n, 1), 17)), np.zeros((max(n, 1), 36))
        lib().orc_slam_get_trajectory(self.h, ptr(rows), ptr(cov), n)
        return rows[:n, :16].reshape(n, 4, 4).copy(), rows[:n, 16].copy(), cov[:n].reshape(n, 6, 6).copy()

    def debug_information(self):
        o = np.zeros(10)
        lib().orc_slam_get_debug_information(self.h, ptr(o))
        return dict(zip(DEBUG_INFORMATION_NAMES, o.tolist()))


class RollingGrid:
    """orc::RollingGrid on its own (RollingGrid.cxx restated), same methods as lidarslam_amd.RollingGrid."""

    def __init__(self, **params):
        self.h = C.c_void_p(lib().orc_grid_create())
        for k, v in params.items():
            self.set(k, v)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_grid_destroy(self.h)
            self.h = None

    def set(self, name, value):
        assert lib().orc_grid_set(self.h, name.encode(), float(value)) == 0, name

    def reset(self, position=None):
        pos = None if position is None else np.ascontiguousarray(position, np.float32)
        lib().orc_grid_reset(self.h, None if pos is None else ptr(pos))

    def clear(self):
        lib().orc_grid_clear(self.h)

    def size(self):
        return lib().orc_grid_size(self.h)

    def roll(self, mn, mx):
        mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
        lib().orc_grid_roll(self.h, ptr(mn), ptr(mx))

    def add(self, pts, fixed=False, time=-1.0, roll=True):
        pts = np.ascontiguousarray(pts, POINT_DTYPE)
        lib().orc_grid_add(self.h, ptr(pts) if pts.size else None, pts.size, int(fixed), float(time), int(roll))

    def clear_old_points(self, time):
        lib().orc_grid_clear_old_points(self.h, float(time))

    def get(self, clean=False):
        out = np.zeros(max(self.size(), 1), POINT_DTYPE)
        n = lib().orc_grid_get(self.h, int(clean), ptr(out), out.size)
        return out[:n].copy()

    def build_submap(self, mn=None, mx=None, min_nb_points=-1):
        if mn is None:
            return lib().orc_grid_build_submap(self.h, None, None, -1)
        mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
        return lib().orc_grid_build_submap(self.h, ptr(mn), ptr(mx), int(min_nb_points))

    def submap_valid(self):
        return bool(lib().orc_grid_submap_valid(self.h))

    def submap(self):
        out = np.zeros(max(self.size(), 1), POINT_DTYPE)
        n = lib().orc_grid_submap(self.h, ptr(out), out.size)
        return out[:n].copy()
