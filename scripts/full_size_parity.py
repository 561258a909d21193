"""Parity at BASELINE.json's full size: a whole synthetic VLS-128 sequence (default 200 frames, ~256 k points each)
through the HIP pipeline -- replayed from the frame store with the look-ahead on, as bench.py does -- and through the
CPU oracle; prints the largest pose difference (reference protocol: translation norm [m] and rotation angle [rad] of
ref^-1 * cur), whether the keypoint sets were identical in every frame, and the keyframe / map sizes at the end.
    python scripts/full_size_parity.py [frames] [model] [seed]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import lidarslam_amd as L  # noqa: E402
from oracle import oracle as O  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
model = int(sys.argv[2]) if len(sys.argv) > 2 else 128
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
L.bind_host_to_device(0)
sg, so = L.Slam(0, EgoMotion=3), O.Slam(EgoMotion=3, NbThreads=16)
scans = [L.synth_frame(model, seed, f) for f in range(frames)]
for f, (pts, _) in enumerate(scans):
    sg.store_frame(f, pts)
worst_t = worst_r = 0.0
same_keypoints = True
t_gpu = t_cpu = 0.0
for f, (pts, stamp) in enumerate(scans):
    if f + 1 < frames:
        sg.hint_next_stored_frame(f + 1)
    t = time.perf_counter(); sg.add_stored_frame(f, stamp, f); t_gpu += time.perf_counter() - t
    t = time.perf_counter(); so.add_frame(pts, stamp, f); t_cpu += time.perf_counter() - t
    D = np.linalg.inv(so.world_transform()) @ sg.world_transform()
    worst_t = max(worst_t, float(np.linalg.norm(D[:3, 3])))
    worst_r = max(worst_r, float(np.arccos(np.clip((np.trace(D[:3, :3]) - 1.0) / 2.0, -1.0, 1.0))))
    if f % 10 == 0 or f == frames - 1:
        for k in (L.EDGE, L.PLANE):
            same_keypoints &= sg.keypoints(k, 2).tobytes() == so.keypoints(k, 2).tobytes()
print(json.dumps({
    "command": "python scripts/full_size_parity.py %d %d %d" % (frames, model, seed), "model": model, "frames": frames, "points_per_frame": int(np.mean([p.size for p, _ in scans])),
    "max_translation_diff_m": worst_t, "max_rotation_diff_rad": worst_r, "keypoint_sets_identical": bool(same_keypoints),
    "keyframes": [sg.stats()[13], so.stats()[13]], "map_sizes_gpu": [int(sg.map(k).size) for k in (L.EDGE, L.PLANE)],
    "map_sizes_oracle": [int(so.map(k).size) for k in (L.EDGE, L.PLANE)],
    "maps_on_device": bool(sg.get_param("DeviceMapsInUse")), "maps_identical_byte_for_byte": bool(all(sg.map(k).tobytes() == so.map(k).tobytes() for k in (L.EDGE, L.PLANE))),
    "sub_maps_identical_byte_for_byte": bool(all(sg.target_submap(k).tobytes() == so.submap(k).tobytes() for k in (L.EDGE, L.PLANE))),
    "solves_that_fell_back_to_the_host_loop": int(sg.get_param("DeviceSolveFallbacks")),
    "gpu_frames_per_s": frames / t_gpu, "oracle_16_threads_frames_per_s": frames / t_cpu,
    "final_position": sg.world_transform()[:3, 3].round(4).tolist()}))
