"""ctypes declarations shared by the product bindings, the tests and bench.py.

Only plain C structs of include/lidarslam_amd.h are mirrored here.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.dirname(os.path.abspath(__file__))
# LSA_LIB: another build of the same ABI, for A/B timing on one box (scripts/ab.sh)
LIB_PATH = os.environ.get("LSA_LIB") or os.path.join(PKG, "liblidarslam_amd.so")
SYNTH_LIB_PATH = os.path.join(PKG, "libslamsynth.so")

# 32-byte LidarPoint (slam_lib/include/LidarSlam/LidarPoint.h:31-64)
POINT_DTYPE = np.dtype(
    [
        ("x", "<f4"),
        ("y", "<f4"),
        ("z", "<f4"),
        ("w", "<f4"),
        ("time", "<f8"),
        ("intensity", "<f4"),
        ("laser_id", "<u2"),
        ("device_id", "u1"),
        ("label", "u1"),
    ],
    align=False,
)
assert POINT_DTYPE.itemsize == 32

EDGE, PLANE, BLOB = 0, 1, 2
SET_RAW_CURRENT, SET_RAW_PREVIOUS, SET_WORKING = 0, 1, 2
MATCH_NSTATUS = 8


class ExtractParams(C.Structure):
    """lsa_extract_params_t with the reference defaults (SSKE.h:125-157)."""

    _fields_ = [
        ("neighbor_width", C.c_int32),
        ("min_distance_to_sensor", C.c_float),
        ("min_beam_surface_angle", C.c_float),
        ("plane_sin_angle_threshold", C.c_float),
        ("edge_sin_angle_threshold", C.c_float),
        ("dist_to_line_threshold", C.c_float),
        ("edge_depth_gap_threshold", C.c_float),
        ("edge_saliency_threshold", C.c_float),
        ("edge_intensity_gap_threshold", C.c_float),
    ]

    def __init__(self, **kw):
        super().__init__(4, 1.5, 10.0, 0.5, 0.86, 0.20, 0.15, 1.5, 50.0)
        for k, v in kw.items():
            setattr(self, k, v)


class MatchParams(C.Structure):
    """lsa_match_params_t with the KeypointsMatcher::Parameters defaults (KeypointsMatcher.h:43-77)."""

    _fields_ = [
        ("single_edge_per_ring", C.c_int32),
        ("edge_nb_neighbors", C.c_int32),
        ("edge_min_nb_neighbors", C.c_int32),
        ("plane_nb_neighbors", C.c_int32),
        ("blob_nb_neighbors", C.c_int32),
        ("reserved", C.c_int32),
        ("max_neighbors_distance", C.c_double),
        ("edge_max_model_error", C.c_double),
        ("planarity_threshold", C.c_double),
        ("plane_max_model_error", C.c_double),
        ("saturation_distance", C.c_double),
    ]

    def __init__(self, **kw):
        super().__init__(0, 10, 4, 5, 10, 0, 5.0, 0.2, 0.04, 0.2, 1.0)
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def ego_motion(cls, **kw):
        """Slam::ComputeEgoMotion's matcher setup (Slam.cxx:877-886, Slam.h:612-628)."""
        return cls(**{**dict(single_edge_per_ring=1, edge_nb_neighbors=8, edge_min_nb_neighbors=3), **kw})

    @classmethod
    def localization(cls, **kw):
        """Slam::Localization's matcher setup (Slam.cxx:1055-1065)."""
        return cls(**{**dict(single_edge_per_ring=0, edge_nb_neighbors=10, edge_min_nb_neighbors=4), **kw})


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint32), ("total_ms", C.c_double), ("bytes", C.c_double)]


def ptr(a, ctype=C.c_void_p):
    """Pointer to a C-contiguous numpy array."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctype)


def pose16(T):
    a = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(16))
    return a


_synth = None


def synth_lib():
    """Host-only generator library (no HIP dependency)."""
    global _synth
    if _synth is None:
        if not os.path.exists(SYNTH_LIB_PATH):
            raise RuntimeError(
                f"{SYNTH_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first"
            )
        lib = C.CDLL(SYNTH_LIB_PATH)
        lib.lsa_synth_frame.restype = C.c_int
        lib.lsa_synth_frame.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        lib.lsa_synth_sensor.restype = C.c_int
        lib.lsa_synth_sensor.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.lsa_synth_pose.restype = None
        lib.lsa_synth_pose.argtypes = [C.c_int, C.c_void_p]
        _synth = lib
    return _synth


def synth_frame(model, seed, frame):
    """One synthetic scan: (points[POINT_DTYPE], stamp_us)."""
    lib = synth_lib()
    nr, nc = C.c_int(), C.c_int()
    e0, e1 = C.c_double(), C.c_double()
    if lib.lsa_synth_sensor(model, nr, nc, e0, e1) != 0:
        raise ValueError(f"unknown sensor model {model}")
    buf = np.zeros(nr.value * nc.value, dtype=POINT_DTYPE)
    stamp = C.c_uint64()
    n = lib.lsa_synth_frame(model, seed, frame, ptr(buf), buf.size, C.byref(stamp))
    if n < 0:
        raise RuntimeError(f"lsa_synth_frame failed: {n}")
    return buf[:n].copy(), int(stamp.value)


def synth_pose(frame):
    T = np.zeros(16)
    synth_lib().lsa_synth_pose(frame, ptr(T))
    return T.reshape(4, 4)
