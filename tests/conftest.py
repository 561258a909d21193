"""Shared fixtures.  `-m "not gpu"` runs on any host (oracle vs golden vectors, host logic, ABI
surface); `-m gpu` are the parity tests proper and call the HIP path through the C ABI.

Only this directory (plus __graft_entry__.smoke and bench.py's cpu_baseline leg) may touch oracle/.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """Build the native pieces once if they are missing (cross-compiles without a GPU)."""
    import lidarslam_amd._native as N

    need = not (os.path.exists(N.LIB_PATH) and os.path.exists(N.SYNTH_LIB_PATH) and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")))
    if need:
        import __graft_entry__ as g

        g.build()


@pytest.fixture(scope="session")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="session")
def L():
    import lidarslam_amd

    return lidarslam_amd


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "mini_seq.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def gpu_ctx(L):
    ctx = L.Context(0)  # raises when there is no device: GPU tests never fall back to the CPU
    yield ctx
    ctx.close()


def pose_diff(ref, cur):
    """The reference's regression protocol (ros_wrapping/tests/src/LidarSlamTestNode.cxx:297-305):
    diff = ref^-1 * cur, translation norm [m] and rotation angle [rad]."""
    D = np.linalg.inv(ref) @ cur
    ang = float(np.arccos(np.clip((np.trace(D[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)))
    return float(np.linalg.norm(D[:3, 3])), ang


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint64) if a.dtype == np.float64 else a


def two_device_rig(L, f, seed=1000, model=8):
    """One synthetic scan split between two LiDAR devices of one platform: even rings stay device 0 (sensor = BASE),
    odd rings become device 1, expressed in that sensor's own frame (BASE <- LIDAR = `offset`), stamped 500 us later
    with the point times shifted to match.  Returns (frames, stamps, offset)."""
    pts, stamp = L.synth_frame(model, seed, f)
    a = np.deg2rad(30.0)
    offset = np.eye(4)
    offset[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    offset[:3, 3] = [-0.4, 0.3, 0.25]
    inv = np.linalg.inv(offset)
    even = pts["laser_id"] % 2 == 0
    d0, d1 = pts[even].copy(), pts[~even].copy()
    d0["laser_id"] //= 2
    d1["laser_id"] //= 2
    xyz = np.stack([d1["x"], d1["y"], d1["z"]], 1).astype(np.float64) @ inv[:3, :3].T + inv[:3, 3]
    d1["x"], d1["y"], d1["z"] = xyz[:, 0].astype(np.float32), xyz[:, 1].astype(np.float32), xyz[:, 2].astype(np.float32)
    d1["device_id"] = 1
    d1["time"] -= 500e-6
    return [d0, d1], [stamp, stamp + 500], offset
