"""The C++ mirror of the reference's public API (lidarslam_amd/include/LidarSlam/Slam.h): a caller
written against LidarSlam::Slam compiles and links against liblidarslam_amd.so (CPU check), and on a
GPU produces the poses of the Python front-end (both sit on the same C ABI)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(tmp_path):
    exe = str(tmp_path / "slam_example")
    pkg = os.path.join(ROOT, "lidarslam_amd")
    cmd = ["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "include"),
           os.path.join(ROOT, "examples", "slam_example.cpp"), "-L" + pkg, "-llidarslam_amd", "-Wl,-rpath," + pkg, "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_caller_compiles_links_and_refuses_to_run_without_a_gpu(tmp_path, L):
    exe = build_example(tmp_path)
    if L.lib().lsa_device_count() == 0:
        r = subprocess.run([exe, "8", "1"], capture_output=True, text=True)
        assert r.returncode == 1 and "no usable HIP device" in r.stderr  # loud failure, no CPU fallback


@pytest.mark.gpu
def test_cpp_caller_matches_the_python_front_end(tmp_path, L):
    exe = build_example(tmp_path)
    r = subprocess.run([exe, "8", "4"], capture_output=True, text=True, check=True)
    rows = np.array([[float(v) for v in line.split()] for line in r.stdout.strip().splitlines()])
    s = L.Slam(0, EgoMotion=3)
    for f in range(4):
        pts, stamp = L.synth_frame(8, 1000, f)
        s.add_frame(pts, stamp, f)
        T = s.world_transform()
        assert np.allclose(rows[f, 1:4], T[:3, 3], atol=1e-11, rtol=0)
        assert int(rows[f, 4]) == s.keypoints(1).size
    s.close()
