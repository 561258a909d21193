"""The C++ mirror of the reference's public API (lidarslam_amd/include/LidarSlam/Slam.h): a caller
written against LidarSlam::Slam compiles and links against liblidarslam_amd.so (CPU check), and on a
GPU produces the poses of the Python front-end (both sit on the same C ABI)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(tmp_path, name="slam_example"):
    exe = str(tmp_path / name)
    pkg = os.path.join(ROOT, "lidarslam_amd")
    cmd = ["g++", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "include"),
           os.path.join(ROOT, "examples", name + ".cpp"), "-L" + pkg, "-llidarslam_amd", "-Wl,-rpath," + pkg, "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_every_call_of_the_reference_wrappers_compiles_against_the_mirror(tmp_path, L):
    """examples/slam_wrapper_calls.cpp makes every call LidarSlamNode.cxx and vtkSlam.cxx make on LidarSlam::Slam (all the
    setters of SetSlamParameters, the getters of PublishOutput, the commands): it compiles and links; without a GPU
    it refuses to run"""
    exe = build_example(tmp_path, "slam_wrapper_calls")
    if L.lib().lsa_device_count() == 0:
        r = subprocess.run([exe, "1"], capture_output=True, text=True)
        assert r.returncode == 1 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
def test_wrapper_calls_run_and_the_out_of_scope_ones_warn(tmp_path, L):
    exe = build_example(tmp_path, "slam_wrapper_calls")
    r = subprocess.run([exe, "3"], capture_output=True, text=True, check=True)
    extra = {line.split()[1]: line.split()[2:] for line in r.stdout.strip().splitlines() if line.startswith("#")}
    assert int(extra["clouds"][3]) == int(extra["clouds"][4]) and int(extra["clouds"][0]) > 0  # registered frame: every point
    assert extra["state"][:4] == ["3", "2", "3", "2"] and extra["sampling"] == ["2", "-1", "odom"]
    assert 0.0 < float(extra["confidence"][0]) <= 1.0 and int(extra["confidence"][2]) > 20
    assert extra["debug"] == ["9", "10", "1"] and extra["cleared"] == ["1", "0"] and extra["reset"] == ["0"]
    for what in ("SaveMapsToPCD", "LoadMapsFromPCD", "RunPoseGraphOptimization", "AddGravityMeasurement", "AddWheelOdomMeasurement"):
        assert f"LidarSlam::Slam::{what}" in r.stderr  # present, warns once, changes nothing


def test_cpp_caller_compiles_links_and_refuses_to_run_without_a_gpu(tmp_path, L):
    exe = build_example(tmp_path)
    if L.lib().lsa_device_count() == 0:
        r = subprocess.run([exe, "8", "1"], capture_output=True, text=True)
        assert r.returncode == 1 and "no usable HIP device" in r.stderr  # loud failure, no CPU fallback


@pytest.mark.gpu
def test_cpp_caller_matches_the_python_front_end(tmp_path, L):
    exe = build_example(tmp_path)
    r = subprocess.run([exe, "8", "4"], capture_output=True, text=True, check=True)
    lines = r.stdout.strip().splitlines()
    rows = np.array([[float(v) for v in line.split()] for line in lines if not line.startswith("#")])
    extra = {line.split()[1]: [float(v) for v in line.split()[2:]] for line in lines if line.startswith("#")}
    s = L.Slam(0, EgoMotion=3, LoggingTimeout=-1, TimeWindowDuration=0.25, VelocityLimitLinear=2.0, VelocityLimitAngular=1000.0)
    for f in range(4):
        pts, stamp = L.synth_frame(8, 1000, f)
        s.add_frame(pts, stamp, f)
        T = s.world_transform()
        assert np.allclose(rows[f, 1:4], T[:3, 3], atol=1e-11, rtol=0)
        assert int(rows[f, 4]) == s.keypoints(1).size
    # the other getters of the C++ mirror against the same calls through the Python front-end
    poses, times, covs = s.trajectory()
    assert extra["trajectory"][:2] == [4, 4] and poses.shape[0] == 4
    assert abs(extra["trajectory"][2] - poses[-1, 0, 3]) < 1e-11
    assert extra["maps"] == [s.map(L.EDGE).size, s.map(L.PLANE).size]
    assert extra["submaps"] == [s.target_submap(L.EDGE).size, s.target_submap(L.PLANE).size]
    info = s.debug_information()
    assert extra["used"] == [info["EgoMotion: edges used"], info["EgoMotion: planes used"], info["Localization: edges used"], info["Localization: planes used"]]
    assert min(extra["used"]) > 20
    assert extra["comply"] == [0.0] == [info["Confidence: comply motion limits"]]
    # the latency differs from run to run: the extrapolation lies ahead of the last pose by (speed x latency)
    ahead, latency = extra["ahead"]
    speed = (poses[-1, 0, 3] - poses[-2, 0, 3]) / (times[-1] - times[-2])
    assert 0 < latency < 0.1 and abs(ahead - (poses[-1, 0, 3] + speed * latency)) < 1e-6
    # the stand-alone extractor on the last frame
    pts, _ = L.synth_frame(8, 1000, 3)
    c = L.Context(0)
    c.upload_frame(pts)
    counts = c.extract_keypoints()
    assert extra["extractor"] == [counts[0], counts[1], c.nb_laser_rings(), pts.size, 10]
    c.close()
    s.close()
    # two devices: the same scan twice, the copy as device 1
    rig = L.Slam(0)
    rig.set_extractor_param(1, "EdgeIntensityGapThreshold", 40.0)
    offset = np.eye(4)
    offset[0, 3] = 0.5
    rig.set_base_to_lidar_offset(offset, 1)
    copy = pts.copy()
    copy["device_id"] = 1
    stamp = L.synth_frame(8, 1000, 3)[1]
    rig.add_frames([pts, copy], [stamp, stamp + 1000], 3)
    assert extra["rig"] == [rig.keypoints(L.EDGE).size, rig.keypoints(L.PLANE).size, 2 * pts.size]
    assert rig.keypoints(L.PLANE).size > counts[1]
    rig.close()
