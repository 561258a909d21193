"""The C-ABI shared library loads and exports every symbol include/lidarslam_amd.h declares
(no compute calls without a GPU), and fails loudly -- never falls back -- when no device exists."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "lidarslam_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lsa_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(L):
    lib = L.lib()
    names = declared_functions()
    assert len(names) > 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_binding_table_matches_header(L):
    assert sorted(L.ABI_SYMBOLS) == declared_functions()


def test_point_layout_is_the_reference_lidarpoint(L):
    # slam_lib/include/LidarSlam/LidarPoint.h:31-64: float[4] @0, double time @16, float intensity @24,
    # u16 laser_id @28, u8 device_id @30, u8 label @31
    d = L.POINT_DTYPE
    assert d.itemsize == 32
    assert [d.fields[n][1] for n in ("x", "y", "z", "w", "time", "intensity", "laser_id", "device_id", "label")] == [0, 4, 8, 12, 16, 24, 28, 30, 31]


def test_param_struct_sizes(L):
    assert C.sizeof(L.ExtractParams) == 36
    assert C.sizeof(L.MatchParams) == 64
    p = L.ExtractParams()
    assert (p.neighbor_width, p.min_distance_to_sensor, p.edge_intensity_gap_threshold) == (4, 1.5, 50.0)
    m = L.MatchParams.ego_motion()
    assert (m.single_edge_per_ring, m.edge_nb_neighbors, m.edge_min_nb_neighbors, m.plane_nb_neighbors) == (1, 8, 3, 5)
    m = L.MatchParams.localization()
    assert (m.single_edge_per_ring, m.edge_nb_neighbors, m.edge_min_nb_neighbors) == (0, 10, 4)


def test_no_cpu_fallback(L):
    """Without a HIP device the product must refuse to run (no silent CPU path)."""
    lib = L.lib()
    if lib.lsa_device_count() > 0:
        return  # on a GPU box this property is exercised by the gpu tests actually running
    h = C.c_void_p()
    assert lib.lsa_ctx_create(0, C.byref(h)) == -1  # LSA_E_NO_DEVICE
    assert not h.value
    s = C.c_void_p()
    assert lib.lsa_slam_create(0, C.byref(s)) == -1
    try:
        L.Slam(0)
    except L.LsaError as e:
        assert "no CPU fallback" in str(e) or "no usable HIP device" in str(e)
    else:
        raise AssertionError("Slam() must raise without a GPU")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lidarslam_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle/" not in txt.replace("oracle/`` is test", "").replace("under ``oracle/``", "") or f == "__init__.py", f
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "orc_" not in txt, f


def test_synthetic_generator_is_deterministic(L):
    a, sa = L.synth_frame(8, 1000, 2)
    b, sb = L.synth_frame(8, 1000, 2)
    c, _ = L.synth_frame(8, 1001, 2)
    assert a.tobytes() == b.tobytes() and sa == sb == 300000
    assert a.tobytes() != c.tobytes()
    # firing order: rings interleaved inside a column, time offsets in [-0.1, 0)
    assert a["time"].min() >= -0.1 and a["time"].max() < 0 and np.all(np.diff(a["time"]) >= 0)
    assert set(np.unique(a["laser_id"])) <= set(range(8))
    T = L.synth_pose(10)
    assert abs(T[0, 3] - 5.0) < 0.01 and np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3))
