"""lidarslam_amd -- MI355X-native scan-matching hot path of Perception4D/LidarSlam.

Python front-end (ctypes) of liblidarslam_amd.so.  Two levels, both thin:

* :class:`Context`  -- the kernel-level C ABI (include/lidarslam_amd.h): upload a scan, extract
  keypoints, set a kNN target, match, accumulate normal equations, undistort.
* :class:`Slam`     -- the pipeline: ``add_frame`` / ``world_transform`` mirror
  ``LidarSlam::Slam::AddFrame`` / ``GetWorldTransform`` (slam_lib/include/LidarSlam/Slam.h:111-146).

There is no CPU fallback: without the built library or without a HIP device every constructor
raises.  The CPU oracle under ``oracle/`` is test infrastructure and is never imported from here.
"""
import ctypes as C
import os

import numpy as np

from ._native import (  # noqa: F401
    BLOB,
    EDGE,
    LIB_PATH,
    MATCH_NSTATUS,
    PLANE,
    POINT_DTYPE,
    SET_RAW_CURRENT,
    SET_RAW_PREVIOUS,
    SET_WORKING,
    ExtractParams,
    KernelStat,
    MatchParams,
    pose16,
    ptr,
    synth_frame,
    synth_pose,
)

__all__ = ["Context", "Slam", "ExtractParams", "MatchParams", "POINT_DTYPE", "lib", "LsaError"]

TARGET_MAP, TARGET_PREVIOUS = 0, 1

DEBUG_INFORMATION_NAMES = [
    "EgoMotion: edges used", "EgoMotion: planes used", "Localization: edges used", "Localization: planes used",
    "Localization: blobs used", "Localization: position error", "Localization: orientation error",
    "Confidence: overlap", "Confidence: comply motion limits", "latency",
]

DEBUG_NAMES = [
    "sin_angle", "saliency", "depth_gap", "intensity_gap", "edge_keypoint", "plane_keypoint", "blob_keypoint",
    "edge_validity", "plane_validity", "blob_validity",
]

# every symbol include/lidarslam_amd.h declares (tests/test_abi.py checks the .so exports them all)
ABI_SYMBOLS = [
    "lsa_device_count", "lsa_bind_host_to_device", "lsa_ctx_create", "lsa_ctx_destroy", "lsa_last_error", "lsa_sync", "lsa_upload_frame", "lsa_upload_wire_frame", "lsa_upload_polydata_frame",
    "lsa_frame_store_put", "lsa_frame_store_use", "lsa_frame_size", "lsa_get_azimuthal_resolution",
    "lsa_set_azimuthal_resolution", "lsa_extract_keypoints", "lsa_extract_keypoints_more", "lsa_extract_prefetch", "lsa_extract_prefetch_adopted", "lsa_transform_frame_at", "lsa_set_keypoint_types", "lsa_download_keypoints", "lsa_keypoint_count",
    "lsa_download_debug", "lsa_nb_laser_rings", "lsa_transform_keypoints", "lsa_set_target", "lsa_set_target_from_set", "lsa_prepare_previous_targets", "lsa_prepared_targets_adopted", "lsa_target_staging", "lsa_set_target_staged", "lsa_stage_target_ahead", "lsa_drop_target_ahead", "lsa_staged_targets_adopted",
    "lsa_target_size", "lsa_download_target", "lsa_set_target_cell_size", "lsa_set_knn_lanes", "lsa_set_fused_match", "lsa_set_knn_rounds", "lsa_match_slow_queries", "lsa_match_exhaustive_queries", "lsa_match_route_stats", "lsa_match_trace", "lsa_set_keypoints", "lsa_match", "lsa_match_types", "lsa_match_types_undistorted",
    "lsa_download_match", "lsa_overlap", "lsa_accumulate", "lsa_mailbox_active", "lsa_solve", "lsa_solve_device", "lsa_solve_device_fallbacks", "lsa_solve_device_begin", "lsa_solve_device_end", "lsa_solve_device_drop", "lsa_icp_gate", "lsa_icp_link", "lsa_solve_device_begin_linked", "lsa_icp_link_peek", "lsa_icp_link_expected", "lsa_icp_post", "lsa_icp_cancel", "lsa_icp_abandon", "lsa_debug_set", "lsa_match_types_gated", "lsa_solve_device_trace", "lsa_registration_error", "lsa_selftest_math", "lsa_reset_working_keypoints", "lsa_undistort", "lsa_working_time_range",
    "lsa_working_bbox", "lsa_working_bboxes", "lsa_localization_begin", "lsa_arm_localization_boxes", "lsa_keypoint_bboxes_begin", "lsa_keypoint_bboxes_begin_interp", "lsa_keypoint_boxes_predicted_mark", "lsa_keypoint_boxes_predicted", "lsa_keypoint_time_range", "lsa_keypoint_bboxes_end", "lsa_download_transformed", "lsa_stage_transformed", "lsa_staged_transformed", "lsa_transform_frame", "lsa_profile_enable", "lsa_profile_select", "lsa_profile_reset",
    "lsa_profile_get", "lsa_slam_create", "lsa_slam_destroy", "lsa_slam_last_error", "lsa_slam_set_param",
    "lsa_slam_get_param", "lsa_slam_reset", "lsa_slam_clear_maps", "lsa_slam_add_frame", "lsa_slam_store_frame", "lsa_slam_add_stored_frame", "lsa_slam_hint_next_stored_frame", "lsa_slam_hint_next_frame", "lsa_upload_frame_begin", "lsa_upload_frame_ready", "lsa_upload_frame_adopt", "lsa_upload_frame_forget", "lsa_profile_event_overhead_us", "lsa_upload_robosense_frame", "lsa_pin_host_memory", "lsa_unpin_host_memory", "lsa_collect_garbage", "lsa_uploads_adopted", "lsa_extract_prefetch_uploaded",
    "lsa_slam_get_world_transform", "lsa_slam_get_covariance", "lsa_slam_get_keypoints", "lsa_slam_get_registered_frame",
    "lsa_slam_get_match_status", "lsa_slam_get_stats", "lsa_slam_context", "lsa_slam_get_latency_compensated_world_transform",
    "lsa_slam_set_world_transform_from_guess", "lsa_slam_get_trajectory", "lsa_slam_get_debug_information", "lsa_slam_get_map",
    "lsa_slam_get_target_submap", "lsa_slam_set_base_to_lidar_offset", "lsa_slam_get_base_to_lidar_offset", "lsa_slam_add_frames", "lsa_slam_set_extractor_param", "lsa_slam_get_extractor_param", "lsa_match_serial", "lsa_match_histogram", "lsa_synth_sensor", "lsa_synth_frame",
    "lsa_synth_pose",
    "lsa_selftest_keep_busy", "lsa_solve_device_interlude", "lsa_device_grid_create", "lsa_device_grid_destroy", "lsa_device_grid_set", "lsa_device_grid_get_param", "lsa_device_grid_reset", "lsa_device_grid_clear",
    "lsa_device_grid_size", "lsa_device_grid_add", "lsa_device_grid_add_keypoints", "lsa_device_grid_roll", "lsa_device_grid_clear_old_points",
    "lsa_device_grid_get", "lsa_device_grid_build_submap", "lsa_device_grid_submap_valid", "lsa_device_grid_stage_keypoints", "lsa_device_grid_stage_keypoints_all", "lsa_device_grid_add_staged", "lsa_device_grid_add_staged_all",
    "lsa_device_grid_build_submap_begin", "lsa_device_grid_build_submap_begin_for_keypoints", "lsa_device_grid_build_submap_end",
    "lsa_device_grid_submap_ahead_begin", "lsa_device_grid_submap_ahead_poll", "lsa_device_grid_submap_ahead_poll_all", "lsa_device_grid_submap_ahead_wait", "lsa_device_grid_submap_ahead_take", "lsa_device_grid_submap_ahead_take_begin", "lsa_device_grid_submap_ahead_take_end",
    "lsa_rolling_grid_create", "lsa_rolling_grid_destroy", "lsa_rolling_grid_set", "lsa_rolling_grid_reset", "lsa_rolling_grid_clear",
    "lsa_rolling_grid_size", "lsa_rolling_grid_roll", "lsa_rolling_grid_add", "lsa_rolling_grid_clear_old_points", "lsa_rolling_grid_get",
    "lsa_rolling_grid_build_submap", "lsa_rolling_grid_submap_valid", "lsa_rolling_grid_submap",
]


class LsaError(RuntimeError):
    pass


class IcpLink(C.Structure):
    """lsa_icp_link_t (include/lidarslam_amd.h): what a solve needs to prepare the ICP iteration enqueued behind it."""

    _fields_ = [
        ("refine_undistortion", C.c_int), ("first", C.c_int), ("have_log", C.c_int),
        ("prev_time", C.c_double), ("cur_time", C.c_double), ("max_extrapolation_ratio", C.c_double),
        ("previous_world", C.c_double * 16), ("motion", C.c_double * 16),
    ]


class SolveResult(C.Structure):
    """lsa_solve_result_t (include/lidarslam_amd.h)."""

    _fields_ = [
        ("pose", C.c_double * 6), ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("cost", C.c_double), ("g", C.c_double * 6), ("H", C.c_double * 36),
        ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int), ("num_iterations", C.c_int),
        ("num_evaluations", C.c_int), ("num_matches", C.c_int), ("skipped", C.c_int), ("termination", C.c_int),
        ("message", C.c_char_p),
    ]


def icp_link_expected(x6, skipped, successful_steps, link):
    """lsa_icp_link_expected: (the 64 words of the block, the motion afterwards) as the host's arithmetic gives them"""
    words, motion = np.zeros(64, np.uint64), np.zeros(16, np.float64)
    x = np.ascontiguousarray(x6, np.float64)
    L = lib()
    L.lsa_icp_link_expected.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    if L.lsa_icp_link_expected(ptr(x), int(skipped), int(successful_steps), C.byref(link), ptr(words), ptr(motion)) != 0:
        raise LsaError("lsa_icp_link_expected")
    return words, motion


_lib = None


def lib():
    """Loads liblidarslam_amd.so (built in-tree by __graft_entry__.build()); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LsaError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
    L.lsa_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.lsa_ctx_destroy.argtypes = [vp]
    L.lsa_last_error.restype = C.c_char_p
    L.lsa_last_error.argtypes = [vp]
    L.lsa_sync.argtypes = [vp]
    L.lsa_upload_frame.argtypes = [vp, vp, i32]
    L.lsa_frame_store_put.argtypes = [vp, i32, vp, i32]
    L.lsa_frame_store_use.argtypes = [vp, i32]
    L.lsa_frame_size.argtypes = [vp]
    L.lsa_get_azimuthal_resolution.restype = C.c_float
    L.lsa_get_azimuthal_resolution.argtypes = [vp]
    L.lsa_set_azimuthal_resolution.argtypes = [vp, C.c_float]
    L.lsa_extract_keypoints.argtypes = [vp, C.POINTER(ExtractParams), vp]
    L.lsa_download_keypoints.argtypes = [vp, i32, i32, vp, i32]
    L.lsa_keypoint_count.argtypes = [vp, i32, i32]
    L.lsa_download_debug.argtypes = [vp, i32, vp, i32]
    L.lsa_nb_laser_rings.argtypes = [vp]
    L.lsa_transform_keypoints.argtypes = [vp, i32, i32, vp, f64]
    L.lsa_set_target.argtypes = [vp, i32, i32, vp, i32]
    L.lsa_set_target_from_set.argtypes = [vp, i32, i32, i32]
    L.lsa_target_size.argtypes = [vp, i32, i32]
    L.lsa_set_target_cell_size.argtypes = [vp, i32, i32, C.c_float]
    L.lsa_match_slow_queries.argtypes = [vp]
    L.lsa_set_keypoints.argtypes = [vp, i32, i32, vp, i32]
    L.lsa_match.argtypes = [vp, i32, i32, i32, C.POINTER(MatchParams), vp, vp]
    L.lsa_set_knn_lanes.argtypes = [vp, i32, i32]
    L.lsa_set_fused_match.argtypes = [vp, i32]
    L.lsa_overlap.argtypes = [vp, C.c_uint, i32, vp, vp, C.c_double, C.c_double, C.c_float, vp, vp]
    L.lsa_upload_wire_frame.argtypes = [vp, vp, i32, vp, vp, i32, i32, C.c_double, i32]
    L.lsa_match_types.argtypes = [vp, i32, C.c_uint, i32, C.POINTER(MatchParams), vp, vp]
    L.lsa_match_types_undistorted.argtypes = [vp, i32, C.c_uint, C.POINTER(MatchParams), vp, vp, vp, vp, f64, f64]
    L.lsa_download_match.argtypes = [vp, i32, vp, vp, vp, i32]
    L.lsa_accumulate.argtypes = [vp, C.c_uint, vp, i32, vp, vp, vp, vp]
    L.lsa_solve.argtypes = [vp, C.c_uint, vp, i32, i32, vp, vp, vp]
    L.lsa_registration_error.argtypes = [vp, C.c_uint, vp, i32, vp, vp]
    L.lsa_solve_device.argtypes = [vp, C.c_uint, vp, i32, i32, i32, C.POINTER(SolveResult)]
    L.lsa_solve_device_fallbacks.argtypes = [vp]
    L.lsa_selftest_math.argtypes = [vp, i32, vp, vp, i32, vp]
    L.lsa_reset_working_keypoints.argtypes = [vp]
    L.lsa_undistort.argtypes = [vp, vp, vp, f64, f64]
    L.lsa_working_time_range.argtypes = [vp, vp, vp]
    L.lsa_keypoint_time_range.argtypes = [vp, i32, vp, vp]
    L.lsa_keypoint_bboxes_begin.argtypes = [vp, i32, vp]
    L.lsa_localization_begin.argtypes = [vp, vp, vp, f64, f64, vp]
    L.lsa_arm_localization_boxes.argtypes = [vp]
    L.lsa_keypoint_bboxes_begin_interp.argtypes = [vp, i32, vp, vp, f64, f64]
    L.lsa_keypoint_bboxes_end.argtypes = [vp, vp, vp]
    L.lsa_keypoint_boxes_predicted_mark.argtypes = [vp]
    L.lsa_keypoint_boxes_predicted.argtypes = [vp, i32, vp, vp, f64, f64]
    L.lsa_working_bbox.argtypes = [vp, i32, vp, vp, vp]
    L.lsa_download_transformed.argtypes = [vp, i32, i32, vp, vp, i32]
    L.lsa_transform_frame.argtypes = [vp, i32, vp, vp, f64, f64, vp, i32]
    L.lsa_profile_enable.argtypes = [vp, i32]
    L.lsa_profile_select.argtypes = [vp, C.c_char_p, i32]
    L.lsa_profile_reset.argtypes = [vp]
    L.lsa_profile_get.argtypes = [vp, vp, i32]
    L.lsa_slam_create.argtypes = [i32, C.POINTER(vp)]
    L.lsa_slam_destroy.argtypes = [vp]
    L.lsa_slam_last_error.restype = C.c_char_p
    L.lsa_slam_last_error.argtypes = [vp]
    L.lsa_slam_set_param.argtypes = [vp, C.c_char_p, f64]
    L.lsa_slam_get_param.argtypes = [vp, C.c_char_p, vp]
    L.lsa_slam_reset.argtypes = [vp, i32]
    L.lsa_slam_add_frame.argtypes = [vp, vp, i32, C.c_uint64, C.c_uint32]
    L.lsa_slam_store_frame.argtypes = [vp, i32, vp, i32]
    L.lsa_slam_add_stored_frame.argtypes = [vp, i32, C.c_uint64, C.c_uint32]
    L.lsa_slam_get_world_transform.argtypes = [vp, vp, vp]
    L.lsa_slam_get_covariance.argtypes = [vp, vp]
    L.lsa_slam_get_keypoints.argtypes = [vp, i32, i32, vp, i32]
    L.lsa_slam_get_registered_frame.argtypes = [vp, vp, i32]
    L.lsa_slam_get_match_status.argtypes = [vp, i32, i32, vp, vp, i32]
    L.lsa_slam_get_stats.argtypes = [vp, vp]
    L.lsa_slam_context.restype = vp
    L.lsa_slam_context.argtypes = [vp]
    L.lsa_slam_get_latency_compensated_world_transform.argtypes = [vp, vp, vp]
    L.lsa_slam_set_world_transform_from_guess.argtypes = [vp, vp]
    L.lsa_slam_get_trajectory.argtypes = [vp, vp, vp, i32]
    L.lsa_slam_hint_next_stored_frame.argtypes = [vp, i32]
    L.lsa_slam_hint_next_frame.argtypes = [vp, vp, i32]
    L.lsa_upload_frame_begin.argtypes = [vp, vp, i32]
    L.lsa_upload_frame_ready.argtypes = [vp]
    L.lsa_upload_frame_adopt.argtypes = [vp, vp, i32]
    L.lsa_uploads_adopted.argtypes = [vp]
    L.lsa_extract_prefetch_uploaded.argtypes = [vp, C.POINTER(ExtractParams)]
    L.lsa_slam_add_frames.argtypes = [vp, vp, vp, vp, vp, i32]
    L.lsa_slam_set_extractor_param.argtypes = [vp, i32, C.c_char_p, f64]
    L.lsa_slam_get_extractor_param.argtypes = [vp, i32, C.c_char_p, vp]
    L.lsa_slam_set_base_to_lidar_offset.argtypes = [vp, vp, i32]
    L.lsa_slam_get_base_to_lidar_offset.argtypes = [vp, vp, i32]
    L.lsa_slam_get_debug_information.argtypes = [vp, vp]
    L.lsa_slam_get_map.argtypes = [vp, i32, i32, vp, i32]
    L.lsa_slam_get_target_submap.argtypes = [vp, i32, vp, i32]
    L.lsa_match_serial.restype = C.c_longlong
    L.lsa_match_serial.argtypes = [vp, i32]
    L.lsa_match_histogram.argtypes = [vp, i32, C.c_longlong, vp]
    L.lsa_rolling_grid_create.restype = vp
    L.lsa_rolling_grid_create.argtypes = []
    L.lsa_rolling_grid_destroy.argtypes = [vp]
    L.lsa_rolling_grid_destroy.restype = None
    L.lsa_rolling_grid_set.argtypes = [vp, C.c_char_p, f64]
    L.lsa_rolling_grid_reset.argtypes = [vp, vp]
    L.lsa_rolling_grid_reset.restype = None
    L.lsa_rolling_grid_clear.argtypes = [vp]
    L.lsa_rolling_grid_clear.restype = None
    L.lsa_rolling_grid_size.argtypes = [vp]
    L.lsa_rolling_grid_roll.argtypes = [vp, vp, vp]
    L.lsa_rolling_grid_roll.restype = None
    L.lsa_rolling_grid_add.argtypes = [vp, vp, i32, i32, f64, i32]
    L.lsa_rolling_grid_clear_old_points.argtypes = [vp, f64]
    L.lsa_rolling_grid_clear_old_points.restype = None
    L.lsa_rolling_grid_get.argtypes = [vp, i32, vp, i32]
    L.lsa_rolling_grid_build_submap.argtypes = [vp, vp, vp, i32]
    L.lsa_rolling_grid_submap_valid.argtypes = [vp]
    L.lsa_rolling_grid_submap.argtypes = [vp, vp, i32]
    _lib = L
    return L


def _profile(L, h):
    buf = (KernelStat * 64)()
    n = L.lsa_profile_get(h, buf, 64)
    return [
        {"name": buf[i].name.decode(), "launches": buf[i].launches, "total_ms": buf[i].total_ms, "bytes": buf[i].bytes}
        for i in range(max(n, 0))
    ]


def bind_host_to_device(device=0):
    """lsa_bind_host_to_device: keep this thread (and the ones created after) on the GPU's NUMA node; returns the
    node, or a negative code when there is nothing to bind to"""
    return lib().lsa_bind_host_to_device(int(device))


class Context:
    """Kernel-level operations of one device context (see include/lidarslam_amd.h)."""

    def __init__(self, device=0, handle=None):
        self.L = lib()
        self._owned = handle is None
        if handle is None:
            h = C.c_void_p()
            rc = self.L.lsa_ctx_create(device, C.byref(h))
            if rc != 0:
                raise LsaError(f"lsa_ctx_create({device}) failed with {rc}: no usable HIP device (no CPU fallback)")
            handle = h
        self.h = handle

    def close(self):
        if getattr(self, "h", None) and self._owned:
            self.L.lsa_ctx_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            raise LsaError(f"{what} failed ({rc}): {self.L.lsa_last_error(self.h).decode()}")
        return rc

    # ---- frame / extraction
    def upload_frame(self, pts):
        pts = np.ascontiguousarray(pts)
        assert pts.dtype == POINT_DTYPE
        self._check(self.L.lsa_upload_frame(self.h, ptr(pts), pts.size), "lsa_upload_frame")

    def store_frame(self, slot, pts):
        pts = np.ascontiguousarray(pts)
        self._check(self.L.lsa_frame_store_put(self.h, slot, ptr(pts), pts.size), "lsa_frame_store_put")

    def use_stored_frame(self, slot):
        self._check(self.L.lsa_frame_store_use(self.h, slot), "lsa_frame_store_use")

    @property
    def azimuthal_resolution(self):
        return self.L.lsa_get_azimuthal_resolution(self.h)

    @azimuthal_resolution.setter
    def azimuthal_resolution(self, v):
        self.L.lsa_set_azimuthal_resolution(self.h, v)

    def upload_wire_frame(self, records, layout, mapping=None, device_id=0, rpm=600.0, timestamp_first_packet=False):
        """lsa_upload_wire_frame: records = contiguous structured / byte array of n driver records;
        layout = (point_step, off_x, off_y, off_z, off_intensity, off_ring, off_time)."""
        rec = np.ascontiguousarray(records)
        lay = (C.c_int32 * 7)(*[int(v) for v in layout])
        n = rec.nbytes // int(layout[0])
        mp = np.ascontiguousarray(mapping, np.uint16) if mapping is not None else None
        self._check(self.L.lsa_upload_wire_frame(self.h, rec.ctypes.data_as(C.c_void_p), n, lay, ptr(mp) if mp is not None else None,
                                                 0 if mp is None else mp.size, device_id, C.c_double(rpm), int(timestamp_first_packet)),
                    "lsa_upload_wire_frame")

    def upload_robosense_frame(self, records, width, height, layout, mapping=None, device_id=0, rpm=600.0):
        """lsa_upload_robosense_frame: the RoboSense driver's organized cloud (height lasers x width points) of records with
        float x, y, z, intensity; layout = (point_step, off_x, off_y, off_z, off_intensity).  Returns the points kept."""
        rec = np.ascontiguousarray(records)
        lay = (C.c_int32 * 7)(*([int(v) for v in layout] + [0, 0]))
        mp = np.ascontiguousarray(mapping, np.uint16) if mapping is not None else None
        kept = C.c_int()
        self.L.lsa_upload_robosense_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        self._check(self.L.lsa_upload_robosense_frame(self.h, rec.ctypes.data_as(C.c_void_p), width, height, lay, ptr(mp) if mp is not None else None,
                                                      0 if mp is None else mp.size, device_id, rpm, C.byref(kept)), "lsa_upload_robosense_frame")
        return kept.value

    def upload_polydata_frame(self, xyz, time, laser_id, intensity, mapping=None, time_to_seconds=1.0):
        """lsa_upload_polydata_frame: the arrays of a vtkPolyData frame (xyz (n, 3) float32 / float64, the others any of
        float32, float64, uint8, uint16, uint32, int32); returns (stamp_us, points kept, all points valid)."""
        codes = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 2, np.dtype(np.uint16): 3, np.dtype(np.uint32): 4, np.dtype(np.int32): 5}
        arrs = [np.ascontiguousarray(a) for a in (xyz, time, laser_id, intensity)]
        mp = np.ascontiguousarray(mapping, np.uint16) if mapping is not None else None
        stamp, kept = C.c_uint64(0), C.c_int(0)
        f = self.L.lsa_upload_polydata_frame
        f.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p, C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        args = []
        for a in arrs:
            args += [a.ctypes.data_as(C.c_void_p), codes[a.dtype]]
        rc = self._check(f(self.h, arrs[1].size, *args, ptr(mp) if mp is not None else None, 0 if mp is None else mp.size, C.c_double(time_to_seconds),
                           C.byref(stamp), C.byref(kept)), "lsa_upload_polydata_frame")
        return stamp.value, kept.value, rc == 1

    def extract_keypoints(self, params=None):
        params = params or ExtractParams()
        counts = np.zeros(3, np.int32)
        self._check(self.L.lsa_extract_keypoints(self.h, C.byref(params), ptr(counts)), "lsa_extract_keypoints")
        return counts

    def keypoints(self, kset, ktype):
        n = self.L.lsa_keypoint_count(self.h, kset, ktype)
        out = np.zeros(max(n, 0), POINT_DTYPE)
        if n > 0:
            self._check(self.L.lsa_download_keypoints(self.h, kset, ktype, ptr(out), n), "lsa_download_keypoints")
        return out

    def set_keypoints(self, kset, ktype, pts):
        pts = np.ascontiguousarray(pts)
        self._check(self.L.lsa_set_keypoints(self.h, kset, ktype, ptr(pts) if pts.size else None, pts.size), "lsa_set_keypoints")

    def debug_array(self, array_id):
        n = self.L.lsa_frame_size(self.h)
        out = np.zeros(n, np.float32)
        self._check(self.L.lsa_download_debug(self.h, array_id, ptr(out), n), "lsa_download_debug")
        return out

    def nb_laser_rings(self):
        return self.L.lsa_nb_laser_rings(self.h)

    def transform_keypoints(self, kset, ktype, T, time_offset=0.0):
        self._check(self.L.lsa_transform_keypoints(self.h, kset, ktype, ptr(pose16(T)), time_offset), "lsa_transform_keypoints")

    # ---- matching
    def set_target(self, ktype, pts, cell=None, slot=TARGET_MAP):
        pts = np.ascontiguousarray(pts)
        if cell is not None:
            self.L.lsa_set_target_cell_size(self.h, slot, ktype, cell)
        self._check(self.L.lsa_set_target(self.h, slot, ktype, ptr(pts) if pts.size else None, pts.size), "lsa_set_target")

    def set_target_from_set(self, ktype, kset, cell=None, slot=TARGET_PREVIOUS):
        if cell is not None:
            self.L.lsa_set_target_cell_size(self.h, slot, ktype, cell)
        self._check(self.L.lsa_set_target_from_set(self.h, slot, ktype, kset), "lsa_set_target_from_set")

    def stage_target(self, ktype, pts, ahead=False, cell=None, slot=TARGET_MAP):
        """lsa_target_staging + lsa_stage_target_ahead / lsa_set_target_staged: the points are written into the
        target's pinned staging buffer, then either handed to the device ahead of time (ahead=True) or made the target"""
        pts = np.ascontiguousarray(pts, POINT_DTYPE)
        self.L.lsa_target_staging.restype = C.c_void_p
        self.L.lsa_target_staging.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        buf = self.L.lsa_target_staging(self.h, slot, ktype, pts.size)
        if not buf and pts.size:
            raise LsaError("lsa_target_staging failed")
        if pts.size:
            C.memmove(buf, pts.ctypes.data, pts.nbytes)
        if cell is not None:
            self.L.lsa_set_target_cell_size(self.h, slot, ktype, cell)
        if ahead:
            self._check(self.L.lsa_stage_target_ahead(self.h, slot, ktype, pts.size), "lsa_stage_target_ahead")
        else:
            self._check(self.L.lsa_set_target_staged(self.h, slot, ktype, pts.size), "lsa_set_target_staged")

    def drop_target_ahead(self, ktype, slot=TARGET_MAP):
        self._check(self.L.lsa_drop_target_ahead(self.h, slot, ktype), "lsa_drop_target_ahead")

    def prepare_previous_targets(self, type_mask):
        self._check(self.L.lsa_prepare_previous_targets(self.h, type_mask), "lsa_prepare_previous_targets")

    def target(self, ktype, slot=TARGET_MAP):
        n = self.L.lsa_target_size(self.h, slot, ktype)
        out = np.zeros(max(n, 0), POINT_DTYPE)
        if n > 0:
            self._check(self.L.lsa_download_target(self.h, slot, ktype, ptr(out), n), "lsa_download_target")
        return out

    def match(self, ktype, query_set, params, pose, slot=TARGET_MAP):
        hist = np.zeros(MATCH_NSTATUS, np.int32)
        self._check(self.L.lsa_match(self.h, slot, ktype, query_set, C.byref(params), ptr(pose16(pose)), ptr(hist)), "lsa_match")
        return hist

    def match_types(self, type_mask, query_set, params, pose, slot=TARGET_MAP, histograms=True):
        """lsa_match_types: the types in type_mask matched concurrently; returns [3][NSTATUS] or None (async)."""
        hist = np.zeros((3, 8), np.int32) if histograms else None
        self._check(self.L.lsa_match_types(self.h, slot, type_mask, query_set, C.byref(params), ptr(pose16(pose)),
                                           ptr(hist) if histograms else None), "lsa_match_types")
        return hist

    def match_types_undistorted(self, type_mask, params, pose, H0, H1, t0, t1, slot=TARGET_MAP):
        """lsa_match_types_undistorted: lsa_undistort + lsa_match_types on the working set, the undistortion inside the search
        kernel where it can be; returns [3][NSTATUS]"""
        hist = np.zeros((3, 8), np.int32)
        self._check(self.L.lsa_match_types_undistorted(self.h, slot, type_mask, C.byref(params), ptr(pose16(pose)), ptr(hist), ptr(pose16(H0)), ptr(pose16(H1)),
                                                       C.c_double(t0), C.c_double(t1)), "lsa_match_types_undistorted")
        return hist

    def overlap(self, type_mask, ratio, leaves, H0, H1=None, t0=0.0, t1=0.0):
        """lsa_overlap: LCP overlap estimate of the current frame against the map-slot targets."""
        out = C.c_float(-1.0)
        lf = (C.c_double * 3)(*[float(x) for x in leaves])
        self._check(self.L.lsa_overlap(self.h, type_mask, int(H1 is not None), ptr(pose16(H0)), ptr(pose16(H1)) if H1 is not None else None,
                                       C.c_double(t0), C.c_double(t1), C.c_float(ratio), lf, C.byref(out)), "lsa_overlap")
        return float(out.value)

    def match_results(self, ktype, query_set=None, records=True):
        n = self.L.lsa_keypoint_count(self.h, SET_WORKING if query_set is None else query_set, ktype)
        n = max(n, 0)
        status = np.zeros(n, np.uint8)
        weights = np.zeros(n, np.float64)
        rec = np.zeros((n, 16), np.float64) if records else None
        got = self._check(
            self.L.lsa_download_match(self.h, ktype, ptr(status), ptr(weights), ptr(rec) if records else None, n), "lsa_download_match"
        )
        return status[:got], weights[:got], (rec[:got] if records else None)

    def route_stats(self, ktype):
        out = np.zeros(8, np.int32)
        self.L.lsa_match_route_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._check(self.L.lsa_match_route_stats(self.h, ktype, ptr(out)), "lsa_match_route_stats")
        return out

    def match_trace(self, blocks):
        out = np.zeros((blocks, 12), np.uint64)
        self.L.lsa_match_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self._check(self.L.lsa_match_trace(self.h, ptr(out), blocks), "lsa_match_trace")
        return out

    def set_fused_match(self, on):
        self._check(self.L.lsa_set_fused_match(self.h, int(on)), "lsa_set_fused_match")

    def slow_queries(self):
        return self.L.lsa_match_slow_queries(self.h)

    def accumulate(self, type_mask, w, jac=True):
        w = np.ascontiguousarray(w, np.float64)
        cost = C.c_double()
        nv = C.c_int()
        g = np.zeros(6)
        H = np.zeros((6, 6))
        self._check(self.L.lsa_accumulate(self.h, type_mask, ptr(w), int(jac), C.byref(cost), ptr(g), ptr(H), C.byref(nv)), "lsa_accumulate")
        return cost.value, g, H, nv.value

    def solve(self, type_mask, prior, max_iter=15, two_d=False):
        """LocalOptimizer::Solve on the device residuals: (pose 4x4, summary[4], costs[2])."""
        out = np.zeros(16)
        summ = np.zeros(4, np.int32)
        costs = np.zeros(2)
        self._check(self.L.lsa_solve(self.h, type_mask, ptr(pose16(prior)), max_iter, int(two_d), ptr(out), ptr(summ), ptr(costs)), "lsa_solve")
        return out.reshape(4, 4), summ, costs

    def solve_device(self, type_mask, prior6, max_iter=15, two_d=False, min_matches=0):
        """LocalOptimizer::Solve as one launch (the trust-region loop runs on the device): SolveResult."""
        r = SolveResult()
        w = np.ascontiguousarray(prior6, np.float64)
        self._check(self.L.lsa_solve_device(self.h, type_mask, ptr(w), int(two_d), max_iter, min_matches, C.byref(r)), "lsa_solve_device")
        return r

    def solve_device_begin(self, type_mask, prior6=None, max_iter=15, two_d=False, min_matches=0):
        """lsa_solve_device_begin; prior6 None: the start point comes from the gate enqueued last (icp_gate)"""
        w = None if prior6 is None else np.ascontiguousarray(prior6, np.float64)
        self.L.lsa_solve_device_begin.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_int, C.c_int, C.c_int]
        self._check(self.L.lsa_solve_device_begin(self.h, type_mask, None if w is None else ptr(w), int(two_d), max_iter, min_matches), "lsa_solve_device_begin")

    def solve_device_end(self):
        """lsa_solve_device_end -> SolveResult, or the negative code (LSA_E_GATE ...) when there is none"""
        r = SolveResult()
        self.L.lsa_solve_device_end.argtypes = [C.c_void_p, C.c_void_p]
        rc = self.L.lsa_solve_device_end(self.h, C.byref(r))
        return r if rc == 0 else rc

    def solve_device_drop(self):
        self.L.lsa_solve_device_drop.argtypes = [C.c_void_p]
        self._check(self.L.lsa_solve_device_drop(self.h), "lsa_solve_device_drop")

    def icp_gate(self):
        self.L.lsa_icp_gate.argtypes = [C.c_void_p]
        return self._check(self.L.lsa_icp_gate(self.h), "lsa_icp_gate")

    def icp_link(self):
        """lsa_icp_link: reserves the block the next linked solve leaves for the iteration behind it -> ticket"""
        self.L.lsa_icp_link.argtypes = [C.c_void_p]
        return self._check(self.L.lsa_icp_link(self.h), "lsa_icp_link")

    def solve_device_begin_linked(self, type_mask, prior6, leave_ticket, link, max_iter=15, two_d=False, min_matches=0):
        w = None if prior6 is None else np.ascontiguousarray(prior6, np.float64)
        self.L.lsa_solve_device_begin_linked.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        self._check(self.L.lsa_solve_device_begin_linked(self.h, type_mask, None if w is None else ptr(w), int(two_d), max_iter, min_matches, leave_ticket,
                                                         None if link is None else C.byref(link)), "lsa_solve_device_begin_linked")

    def icp_link_peek(self, ticket):
        out = np.zeros(64, np.uint64)
        self.L.lsa_icp_link_peek.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._check(self.L.lsa_icp_link_peek(self.h, ticket, ptr(out)), "lsa_icp_link_peek")
        return out

    def icp_post(self, ticket, pose, prior6, H0=None, H1=None, t0=0.0, t1=0.0):
        w = np.ascontiguousarray(prior6, np.float64)
        self.L.lsa_icp_post.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
        self._check(self.L.lsa_icp_post(self.h, ticket, ptr(pose16(pose)), ptr(w), None if H0 is None else ptr(pose16(H0)), None if H1 is None else ptr(pose16(H1)), t0, t1), "lsa_icp_post")

    def icp_cancel(self, ticket):
        self.L.lsa_icp_cancel.argtypes = [C.c_void_p, C.c_int]
        self._check(self.L.lsa_icp_cancel(self.h, ticket), "lsa_icp_cancel")

    def icp_abandon(self):
        self.L.lsa_icp_abandon.argtypes = [C.c_void_p]
        self._check(self.L.lsa_icp_abandon(self.h), "lsa_icp_abandon")

    def match_types_gated(self, type_mask, query_set, params, undistort=False, slot=TARGET_MAP):
        """lsa_match_types_gated: 0 enqueued behind the gate, 1 this match cannot wait behind a gate"""
        self.L.lsa_match_types_gated.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_void_p, C.c_int]
        return self._check(self.L.lsa_match_types_gated(self.h, slot, type_mask, query_set, C.byref(params), int(undistort)), "lsa_match_types_gated")

    def profile_event_overhead_us(self):
        self.L.lsa_profile_event_overhead_us.restype = C.c_double
        self.L.lsa_profile_event_overhead_us.argtypes = [C.c_void_p]
        return float(self.L.lsa_profile_event_overhead_us(self.h))

    def debug_set(self, name, value):
        self.L.lsa_debug_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        self._check(self.L.lsa_debug_set(self.h, name.encode(), int(value)), "lsa_debug_set")

    def solve_device_trace(self):
        out = np.zeros(12, np.uint64)
        self.L.lsa_solve_device_trace.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.L.lsa_solve_device_trace(self.h, ptr(out)), "lsa_solve_device_trace")
        return out

    def solve_device_fallbacks(self):
        return self.L.lsa_solve_device_fallbacks(self.h)

    def registration_error(self, type_mask, pose, two_d=False):
        cov = np.zeros((6, 6))
        err = np.zeros(2)
        self._check(self.L.lsa_registration_error(self.h, type_mask, ptr(pose16(pose)), int(two_d), ptr(cov), ptr(err)), "lsa_registration_error")
        return cov, err

    def selftest_math(self, fn, x, y=None):
        x = np.ascontiguousarray(x, np.float64)
        y = np.ascontiguousarray(x if y is None else y, np.float64)
        out = np.zeros_like(x)
        self._check(self.L.lsa_selftest_math(self.h, fn, ptr(x), ptr(y), x.size, ptr(out)), "lsa_selftest_math")
        return out

    # ---- undistortion / transforms
    def reset_working_keypoints(self):
        self._check(self.L.lsa_reset_working_keypoints(self.h), "lsa_reset_working_keypoints")

    def undistort(self, H0, H1, t0, t1):
        self._check(self.L.lsa_undistort(self.h, ptr(pose16(H0)), ptr(pose16(H1)), t0, t1), "lsa_undistort")

    def localization_begin(self, H0=None, H1=None, t0=0.0, t1=0.0, box_pose=None, arm=False):
        """lsa_localization_begin: reset + undistort (unless H0 is None) + boxes under box_pose (unless None), one launch;
        the boxes are read with keypoint_bboxes_end"""
        if arm:
            self._check(self.L.lsa_arm_localization_boxes(self.h), "lsa_arm_localization_boxes")
        self._check(self.L.lsa_localization_begin(self.h, ptr(pose16(H0)) if H0 is not None else None, ptr(pose16(H1)) if H1 is not None else None,
                                                  C.c_double(t0), C.c_double(t1), ptr(pose16(box_pose)) if box_pose is not None else None), "lsa_localization_begin")

    def keypoint_bboxes_end(self):
        mn, mx = np.zeros(9, np.float32), np.zeros(9, np.float32)
        self._check(self.L.lsa_keypoint_bboxes_end(self.h, ptr(mn), ptr(mx)), "lsa_keypoint_bboxes_end")
        return mn.reshape(3, 3), mx.reshape(3, 3)

    def working_time_range(self):
        a, b = C.c_double(), C.c_double()
        self._check(self.L.lsa_working_time_range(self.h, C.byref(a), C.byref(b)), "lsa_working_time_range")
        return a.value, b.value

    def keypoint_time_range(self, kset):
        a, b = C.c_double(), C.c_double()
        self._check(self.L.lsa_keypoint_time_range(self.h, kset, C.byref(a), C.byref(b)), "lsa_keypoint_time_range")
        return a.value, b.value

    def keypoint_bboxes(self, kset, H0, H1=None, t0=0.0, t1=0.0):
        """bounding boxes of the three keypoint types of a set under a pose, or under the pose interpolated at
        every point's time between H0 (t0) and H1 (t1); returns (mn[3][3], mx[3][3])"""
        if H1 is None:
            self._check(self.L.lsa_keypoint_bboxes_begin(self.h, kset, ptr(pose16(H0))), "lsa_keypoint_bboxes_begin")
        else:
            self._check(self.L.lsa_keypoint_bboxes_begin_interp(self.h, kset, ptr(pose16(H0)), ptr(pose16(H1)), C.c_double(t0), C.c_double(t1)),
                        "lsa_keypoint_bboxes_begin_interp")
        mn, mx = np.zeros(9, np.float32), np.zeros(9, np.float32)
        self._check(self.L.lsa_keypoint_bboxes_end(self.h, ptr(mn), ptr(mx)), "lsa_keypoint_bboxes_end")
        return mn.reshape(3, 3), mx.reshape(3, 3)

    def match_serial(self, ktype):
        return self.L.lsa_match_serial(self.h, ktype)

    def match_histogram(self, ktype, serial):
        h = np.zeros(8, np.int32)
        self._check(self.L.lsa_match_histogram(self.h, ktype, serial, ptr(h)), "lsa_match_histogram")
        return h

    def working_bbox(self, ktype, pose):
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._check(self.L.lsa_working_bbox(self.h, ktype, ptr(pose16(pose)), ptr(mn), ptr(mx)), "lsa_working_bbox")
        return mn, mx

    def transformed_keypoints(self, kset, ktype, pose):
        n = max(self.L.lsa_keypoint_count(self.h, kset, ktype), 0)
        out = np.zeros(n, POINT_DTYPE)
        if n:
            self._check(self.L.lsa_download_transformed(self.h, kset, ktype, ptr(pose16(pose)), ptr(out), n), "lsa_download_transformed")
        return out

    def transform_frame(self, H0, H1=None, t0=0.0, t1=0.0):
        n = self.L.lsa_frame_size(self.h)
        out = np.zeros(n, POINT_DTYPE)
        interp = H1 is not None
        self._check(
            self.L.lsa_transform_frame(self.h, int(interp), ptr(pose16(H0)), ptr(pose16(H1)) if interp else None, t0, t1, ptr(out), n),
            "lsa_transform_frame",
        )
        return out

    # ---- profiling
    def profile(self, on=True):
        self.L.lsa_profile_enable(self.h, int(on))

    def profile_select(self, scope, every=1):
        self.L.lsa_profile_select(self.h, scope.encode(), int(every))

    def profile_reset(self):
        self.L.lsa_profile_reset(self.h)

    def profile_stats(self):
        return _profile(self.L, self.h)

    def sync(self):
        self._check(self.L.lsa_sync(self.h), "lsa_sync")


class Slam:
    """LidarSlam::Slam on one MI355X.  Parameters use the reference's names (``EgoMotion=3`` ...)."""

    def __init__(self, device=0, **params):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.lsa_slam_create(device, C.byref(h))
        if rc != 0:
            raise LsaError(f"lsa_slam_create({device}) failed with {rc}: no usable HIP device (there is no CPU fallback)")
        self.h = h
        self._n = 0
        for k, v in params.items():
            self.set_param(k, v)

    def close(self):
        if getattr(self, "h", None):
            self.L.lsa_slam_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            raise LsaError(f"{what} failed ({rc}): {self.L.lsa_slam_last_error(self.h).decode()}")
        return rc

    def set_param(self, name, value):
        if self.L.lsa_slam_set_param(self.h, name.encode(), float(value)) != 0:
            raise KeyError(name)

    def get_param(self, name):
        v = C.c_double()
        if self.L.lsa_slam_get_param(self.h, name.encode(), C.byref(v)) != 0:
            raise KeyError(name)
        return v.value

    def reset(self, reset_log=True):
        self.L.lsa_slam_reset(self.h, int(reset_log))

    def add_frame(self, pts, stamp_us, seq=0):
        """Slam::AddFrame: host scan -> pose."""
        pts = np.ascontiguousarray(pts)
        assert pts.dtype == POINT_DTYPE
        self._n = pts.size
        self._check(self.L.lsa_slam_add_frame(self.h, ptr(pts), pts.size, stamp_us, seq), "lsa_slam_add_frame")

    def store_frame(self, slot, pts):
        pts = np.ascontiguousarray(pts)
        self._check(self.L.lsa_slam_store_frame(self.h, slot, ptr(pts), pts.size), "lsa_slam_store_frame")

    def hint_next_stored_frame(self, slot):
        """replay: the slot that will be added after the next add_stored_frame (its extraction is overlapped)"""
        self._check(self.L.lsa_slam_hint_next_stored_frame(self.h, slot), "lsa_slam_hint_next_stored_frame")

    def hint_next_frame(self, pts):
        """replay from host clouds: the array of the add_frame call after the next one; its upload and extraction overlap
        the frame in between.  The very same (contiguous) array must then be passed to add_frame and stay alive until then."""
        self._check(self.L.lsa_slam_hint_next_frame(self.h, ptr(pts), pts.size), "lsa_slam_hint_next_frame")

    def add_stored_frame(self, slot, stamp_us, seq=0):
        self._check(self.L.lsa_slam_add_stored_frame(self.h, slot, stamp_us, seq), "lsa_slam_add_stored_frame")

    # replay loops that hold their clouds for the whole run: the pointer of a cloud is taken once (`cloud_pointer`), not per call
    @staticmethod
    def pin_cloud(pts):
        """lsa_pin_host_memory on the array's buffer (a driver's scan buffer that is handed over again and again)"""
        L = lib()
        L.lsa_pin_host_memory.argtypes = [C.c_void_p, C.c_size_t]
        if L.lsa_pin_host_memory(ptr(pts), pts.nbytes) != 0:
            raise LsaError("lsa_pin_host_memory failed")

    @staticmethod
    def unpin_cloud(pts):
        L = lib()
        L.lsa_unpin_host_memory.argtypes = [C.c_void_p]
        L.lsa_unpin_host_memory(ptr(pts))

    @staticmethod
    def cloud_pointer(pts):
        assert pts.dtype == POINT_DTYPE
        return ptr(pts), int(pts.size)

    def add_frame_at(self, cloud, stamp_us, seq=0):
        """add_frame on a (pointer, size) pair from `cloud_pointer`; the array it came from must be alive"""
        self._n = cloud[1]
        self._check(self.L.lsa_slam_add_frame(self.h, cloud[0], cloud[1], stamp_us, seq), "lsa_slam_add_frame")

    def hint_next_frame_at(self, cloud):
        self._check(self.L.lsa_slam_hint_next_frame(self.h, cloud[0], cloud[1]), "lsa_slam_hint_next_frame")

    def stats_into(self, out_ptr):
        """stats() into an array the caller keeps (out_ptr = ptr(np.zeros(16)))"""
        self.L.lsa_slam_get_stats(self.h, out_ptr)

    def world_transform(self):
        T = np.zeros(16)
        t = C.c_double()
        self.L.lsa_slam_get_world_transform(self.h, ptr(T), C.byref(t))
        return T.reshape(4, 4)

    def covariance(self):
        c = np.zeros((6, 6))
        self.L.lsa_slam_get_covariance(self.h, ptr(c))
        return c

    def latency_compensated_world_transform(self):
        T = np.zeros(16)
        t = C.c_double()
        self._check(self.L.lsa_slam_get_latency_compensated_world_transform(self.h, ptr(T), C.byref(t)), "latency compensated transform")
        return T.reshape(4, 4)

    def set_world_transform_from_guess(self, T):
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        self._check(self.L.lsa_slam_set_world_transform_from_guess(self.h, ptr(T)), "lsa_slam_set_world_transform_from_guess")

    def add_frames(self, frames, stamps_us, seq=0):
        """Slam::AddFrames: one frame per LiDAR device, each with its own stamp"""
        frames = [np.ascontiguousarray(f, POINT_DTYPE) for f in frames]
        ptrs = (C.c_void_p * len(frames))(*[f.ctypes.data if f.size else None for f in frames])
        sizes = np.array([f.size for f in frames], np.int32)
        stamps = np.array(stamps_us, np.uint64)
        seqs = np.full(len(frames), seq, np.uint32)
        self._n = max(self._n, int(sizes.sum()))
        return self._check(self.L.lsa_slam_add_frames(self.h, ptrs, ptr(sizes), ptr(stamps), ptr(seqs), len(frames)), "lsa_slam_add_frames")

    def set_extractor_param(self, device_id, name, value):
        self._check(self.L.lsa_slam_set_extractor_param(self.h, device_id, name.encode(), float(value)), "lsa_slam_set_extractor_param")

    def extractor_param(self, device_id, name):
        v = C.c_double()
        self._check(self.L.lsa_slam_get_extractor_param(self.h, device_id, name.encode(), C.byref(v)), "lsa_slam_get_extractor_param")
        return v.value

    def set_base_to_lidar_offset(self, T, device_id=0):
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        self._check(self.L.lsa_slam_set_base_to_lidar_offset(self.h, ptr(T), device_id), "lsa_slam_set_base_to_lidar_offset")

    def base_to_lidar_offset(self, device_id=0):
        T = np.zeros(16)
        self._check(self.L.lsa_slam_get_base_to_lidar_offset(self.h, ptr(T), device_id), "lsa_slam_get_base_to_lidar_offset")
        return T.reshape(4, 4)

    def trajectory(self):
        """Slam::GetTrajectory / GetCovariances: (n, 4, 4) poses, (n,) times [s], (n, 6, 6) covariances"""
        n = self._check(self.L.lsa_slam_get_trajectory(self.h, None, None, 0), "lsa_slam_get_trajectory")
        rows, cov = np.zeros((max(n, 1), 17)), np.zeros((max(n, 1), 36))
        self.L.lsa_slam_get_trajectory(self.h, ptr(rows), ptr(cov), n)
        return rows[:n, :16].reshape(n, 4, 4).copy(), rows[:n, 16].copy(), cov[:n].reshape(n, 6, 6).copy()

    def debug_information(self):
        """Slam::GetDebugInformation with the reference's keys (+ "latency")"""
        o = np.zeros(10)
        self._check(self.L.lsa_slam_get_debug_information(self.h, ptr(o)), "lsa_slam_get_debug_information")
        return dict(zip(DEBUG_INFORMATION_NAMES, o.tolist()))

    def map(self, ktype, clean=False):
        """Slam::GetMap(k, clean)"""
        n = self._check(self.L.lsa_slam_get_map(self.h, ktype, int(clean), None, 0), "lsa_slam_get_map")
        out = np.zeros(max(n, 1), POINT_DTYPE)
        n = self.L.lsa_slam_get_map(self.h, ktype, int(clean), ptr(out), out.size)
        return out[:n].copy()

    def target_submap(self, ktype):
        """Slam::GetTargetSubMap(k)"""
        n = self._check(self.L.lsa_slam_get_target_submap(self.h, ktype, None, 0), "lsa_slam_get_target_submap")
        out = np.zeros(max(n, 1), POINT_DTYPE)
        n = self.L.lsa_slam_get_target_submap(self.h, ktype, ptr(out), out.size)
        return out[:n].copy()

    def keypoints(self, ktype, which=0, cap=400000):
        """which: 0 undistorted BASE, 1 WORLD, 2 raw BASE."""
        out = np.zeros(cap, POINT_DTYPE)
        n = self._check(self.L.lsa_slam_get_keypoints(self.h, ktype, which, ptr(out), cap), "lsa_slam_get_keypoints")
        return out[:n].copy()

    def registered_frame(self, cap=None):
        cap = cap or max(self._n, 1 << 19)
        out = np.zeros(cap, POINT_DTYPE)
        n = self._check(self.L.lsa_slam_get_registered_frame(self.h, ptr(out), cap), "lsa_slam_get_registered_frame")
        return out[:n]

    def match_status(self, localization, ktype, cap=400000):
        st = np.zeros(cap, np.uint8)
        w = np.zeros(cap)
        n = self.L.lsa_slam_get_match_status(self.h, int(localization), ktype, ptr(st), ptr(w), cap)
        return st[:n].copy(), w[:n].copy()

    def stats(self):
        o = np.zeros(16)
        self.L.lsa_slam_get_stats(self.h, ptr(o))
        return o

    def context(self):
        return Context(handle=C.c_void_p(self.L.lsa_slam_context(self.h)))


class RollingGrid:
    """LidarSlam::RollingGrid (RollingGrid.h:63-212) as the pipeline uses it: host code, no device involved."""

    def __init__(self, **params):
        self.L = lib()
        self.h = C.c_void_p(self.L.lsa_rolling_grid_create())
        if not self.h:
            raise LsaError("lsa_rolling_grid_create failed")
        for k, v in params.items():
            self.set(k, v)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.lsa_rolling_grid_destroy(self.h)
            self.h = None

    def set(self, name, value):
        if self.L.lsa_rolling_grid_set(self.h, name.encode(), float(value)) != 0:
            raise LsaError(f"lsa_rolling_grid_set({name}, {value}) refused")

    def reset(self, position=None):
        pos = None if position is None else np.ascontiguousarray(position, np.float32)
        self.L.lsa_rolling_grid_reset(self.h, None if pos is None else ptr(pos))

    def clear(self):
        self.L.lsa_rolling_grid_clear(self.h)

    def size(self):
        return self.L.lsa_rolling_grid_size(self.h)

    def roll(self, mn, mx):
        mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
        self.L.lsa_rolling_grid_roll(self.h, ptr(mn), ptr(mx))

    def add(self, pts, fixed=False, time=-1.0, roll=True):
        pts = np.ascontiguousarray(pts, POINT_DTYPE)
        if self.L.lsa_rolling_grid_add(self.h, ptr(pts) if pts.size else None, pts.size, int(fixed), float(time), int(roll)) != 0:
            raise LsaError("lsa_rolling_grid_add failed")

    def clear_old_points(self, time):
        self.L.lsa_rolling_grid_clear_old_points(self.h, float(time))

    def get(self, clean=False):
        out = np.zeros(max(self.size(), 1), POINT_DTYPE)
        n = self.L.lsa_rolling_grid_get(self.h, int(clean), ptr(out), out.size)
        return out[:n].copy()

    def build_submap(self, mn=None, mx=None, min_nb_points=-1):
        if mn is None:
            return self.L.lsa_rolling_grid_build_submap(self.h, None, None, -1)
        mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
        return self.L.lsa_rolling_grid_build_submap(self.h, ptr(mn), ptr(mx), int(min_nb_points))

    def submap_valid(self):
        return bool(self.L.lsa_rolling_grid_submap_valid(self.h))

    def submap(self):
        out = np.zeros(max(self.size(), 1), POINT_DTYPE)
        n = self.L.lsa_rolling_grid_submap(self.h, ptr(out), out.size)
        return out[:n].copy()


class DeviceGrid:
    """LidarSlam::RollingGrid resident on the device (lsa_device_grid_*): same calls as RollingGrid, the map lives in the
    memory of `ctx` and its sub-map becomes a kNN target of that context."""

    def __init__(self, ctx, **params):
        self.L = lib()
        self.ctx = ctx
        h = C.c_void_p()
        vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
        self.L.lsa_device_grid_create.argtypes = [vp, C.POINTER(vp)]
        self.L.lsa_device_grid_destroy.argtypes = [vp]
        self.L.lsa_device_grid_set.argtypes = [vp, C.c_char_p, f64]
        self.L.lsa_device_grid_reset.argtypes = [vp, vp]
        self.L.lsa_device_grid_clear.argtypes = [vp]
        self.L.lsa_device_grid_size.argtypes = [vp]
        self.L.lsa_device_grid_add.argtypes = [vp, vp, i32, i32, f64, i32]
        self.L.lsa_device_grid_add_keypoints.argtypes = [vp, i32, i32, vp, f64]
        self.L.lsa_device_grid_roll.argtypes = [vp, vp, vp]
        self.L.lsa_device_grid_clear_old_points.argtypes = [vp, f64]
        self.L.lsa_device_grid_get.argtypes = [vp, i32, vp, i32]
        self.L.lsa_device_grid_build_submap.argtypes = [vp, vp, vp, i32, i32, i32]
        self.L.lsa_device_grid_submap_valid.argtypes = [vp]
        self.L.lsa_device_grid_stage_keypoints.argtypes = [vp, i32, i32, vp]
        self.L.lsa_device_grid_add_staged.argtypes = [vp, f64]
        self.L.lsa_device_grid_build_submap_begin.argtypes = [vp, vp, vp, i32, i32, i32]
        self.L.lsa_device_grid_build_submap_begin_for_keypoints.argtypes = [vp, i32, i32, i32, i32]
        self.L.lsa_device_grid_build_submap_end.argtypes = [vp]
        if self.L.lsa_device_grid_create(ctx.h, C.byref(h)) != 0:
            raise LsaError("lsa_device_grid_create failed")
        self.h = h
        for k, v in params.items():
            self.set(k, v)

    def close(self):
        if getattr(self, "h", None):
            if self.ctx.h:  # a grid must go before its context (include/lidarslam_amd.h); one that outlived it is dropped
                self.L.lsa_device_grid_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            raise LsaError(f"{what} failed ({rc}): {self.L.lsa_last_error(self.ctx.h).decode()}")
        return rc

    def set(self, name, value):
        self._check(self.L.lsa_device_grid_set(self.h, name.encode(), float(value)), f"lsa_device_grid_set({name})")

    def reset(self, position=None):
        pos = None if position is None else np.ascontiguousarray(position, np.float32)
        self._check(self.L.lsa_device_grid_reset(self.h, None if pos is None else ptr(pos)), "lsa_device_grid_reset")

    def clear(self):
        self._check(self.L.lsa_device_grid_clear(self.h), "lsa_device_grid_clear")

    def size(self):
        return self._check(self.L.lsa_device_grid_size(self.h), "lsa_device_grid_size")

    def roll(self, mn, mx):
        mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
        self._check(self.L.lsa_device_grid_roll(self.h, ptr(mn), ptr(mx)), "lsa_device_grid_roll")

    def add(self, pts, fixed=False, time=-1.0, roll=True):
        pts = np.ascontiguousarray(pts, POINT_DTYPE)
        self._check(self.L.lsa_device_grid_add(self.h, ptr(pts) if pts.size else None, pts.size, int(fixed), float(time), int(roll)), "lsa_device_grid_add")

    def add_keypoints(self, kset, ktype, pose, time):
        self._check(self.L.lsa_device_grid_add_keypoints(self.h, kset, ktype, ptr(pose16(pose)), float(time)), "lsa_device_grid_add_keypoints")

    def stage_keypoints(self, kset, ktype, pose):
        self._check(self.L.lsa_device_grid_stage_keypoints(self.h, kset, ktype, ptr(pose16(pose))), "lsa_device_grid_stage_keypoints")

    def add_staged(self, time):
        """the insertion of what stage_keypoints read; any host thread"""
        self._check(self.L.lsa_device_grid_add_staged(self.h, float(time)), "lsa_device_grid_add_staged")

    def build_submap_begin_for_keypoints(self, box_type, min_nb_points, ktype=PLANE, slot=TARGET_MAP):
        """box of the keypoints of `box_type` as Context.keypoint_bboxes_begin left it on the device"""
        self._check(self.L.lsa_device_grid_build_submap_begin_for_keypoints(self.h, box_type, int(min_nb_points), slot, ktype), "lsa_device_grid_build_submap_begin_for_keypoints")

    def build_submap_end(self):
        return self._check(self.L.lsa_device_grid_build_submap_end(self.h), "lsa_device_grid_build_submap_end")

    def clear_old_points(self, time):
        self._check(self.L.lsa_device_grid_clear_old_points(self.h, float(time)), "lsa_device_grid_clear_old_points")

    def get(self, clean=False, capacity=1 << 22):
        out = np.zeros(capacity, POINT_DTYPE)
        n = self._check(self.L.lsa_device_grid_get(self.h, int(clean), ptr(out), out.size), "lsa_device_grid_get")
        return out[:n].copy()

    def build_submap(self, mn=None, mx=None, min_nb_points=-1, ktype=PLANE, slot=TARGET_MAP):
        """the sub-map becomes target (slot, ktype) of the context; returns its size"""
        if mn is None:
            return self._check(self.L.lsa_device_grid_build_submap(self.h, None, None, -1, slot, ktype), "lsa_device_grid_build_submap")
        mn, mx = np.ascontiguousarray(mn, np.float32), np.ascontiguousarray(mx, np.float32)
        return self._check(self.L.lsa_device_grid_build_submap(self.h, ptr(mn), ptr(mx), int(min_nb_points), slot, ktype), "lsa_device_grid_build_submap")

    def submap_valid(self):
        return bool(self.L.lsa_device_grid_submap_valid(self.h))
