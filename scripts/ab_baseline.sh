#!/bin/bash
# builds HEAD's library into lidarslam_amd/_ab/ (working tree changes are stashed and restored)
set -e
cd "$(dirname "$0")/.."
git stash -q
make -C lidarslam_amd/csrc -j8 > /dev/null
mkdir -p lidarslam_amd/_ab && cp lidarslam_amd/liblidarslam_amd.so lidarslam_amd/_ab/
git stash pop -q
touch lidarslam_amd/csrc/*.hip lidarslam_amd/csrc/host/*.cpp
make -C lidarslam_amd/csrc -j8 > /dev/null
