"""Sanitizers on the CPU build of the host code that runs on several threads (GPU sanitizers are not available on the
pool): the rolling grid's parallel Add / BuildSubMap under ThreadSanitizer and Address + UB sanitizer, compared with
the single-threaded map byte for byte by the driver itself."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", ["thread", "address,undefined"])
def test_parallel_rolling_grid_is_clean_under_sanitizers(tmp_path, flags):
    exe = str(tmp_path / "drv")
    host = os.path.join(ROOT, "lidarslam_amd", "csrc", "host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + flags, "-I" + host, "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "rolling_grid_sanitize.cpp"), os.path.join(host, "lsa_rolling_grid.cpp"), "-lpthread", "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this compiler has no runtime for -fsanitize=" + flags)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.startswith("ok"), (run.stdout[-500:], run.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in run.stderr and "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
