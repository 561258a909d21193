/*
 * lidarslam_amd.h -- C ABI of liblidarslam_amd.so: the MI355X-native
 * implementation of the per-frame scan-matching hot path of
 * Perception4D/LidarSlam (slam_lib).
 *
 * The reference has no FFI layer: the path sits behind the C++ API of
 * libLidarSlam.  This header is the seam a maintainer binds
 * instead of the four internal call sites listed below.  Plain pointers and
 * sizes only -- no STL / Eigen / PCL / torch types cross this boundary.  All
 * device memory is owned by the context; the host never sees device pointers.
 *
 * All functions return 0 on success and a negative LSA_E_* code on failure;
 * lsa_last_error() returns a human readable message for the last failure of
 * that context.  There is NO CPU fallback: when no HIP device is usable,
 * lsa_ctx_create fails with LSA_E_NO_DEVICE and nothing else can be called.
 *
 * file:line citations are relative to /root/reference.
 */
#ifndef LIDARSLAM_AMD_H
#define LIDARSLAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSA_OK 0
#define LSA_E_NO_DEVICE (-1)
#define LSA_E_HIP (-2)
#define LSA_E_ARG (-3)
#define LSA_E_STATE (-4)
#define LSA_E_CAPACITY (-5)
#define LSA_E_GATE (-6) /* an ICP iteration enqueued ahead did not run: its gate gave up waiting for the host (lsa_icp_gate) */

/* Keypoint types -- slam_lib/include/LidarSlam/Enums.h:30-36 */
#define LSA_EDGE 0
#define LSA_PLANE 1
#define LSA_BLOB 2

/* The 32-byte PCL point of the reference, byte for byte
 * -- slam_lib/include/LidarSlam/LidarPoint.h:31-64 (registered :68-77). */
typedef struct lsa_point_t
{
  float x, y, z, w;   /* PCL_ADD_POINT4D: data[4], w == 1 */
  double time;        /* offset in seconds to the frame stamp */
  float intensity;
  uint16_t laser_id;  /* ring, 0 = lowest */
  uint8_t device_id;
  uint8_t label;
} lsa_point_t;

/* Match status values -- slam_lib/include/LidarSlam/KeypointsMatcher.h:82-93 */
enum
{
  LSA_MATCH_SUCCESS = 0,
  LSA_MATCH_BAD_MODEL_PARAMETRIZATION = 1,
  LSA_MATCH_NOT_ENOUGH_NEIGHBORS = 2,
  LSA_MATCH_NEIGHBORS_TOO_FAR = 3,
  LSA_MATCH_BAD_PCA_STRUCTURE = 4,
  LSA_MATCH_INVALID_NUMERICAL = 5,
  LSA_MATCH_MSE_TOO_LARGE = 6,
  LSA_MATCH_UNKOWN = 7,
  LSA_MATCH_NSTATUS = 8
};

/* Parameters of SpinningSensorKeypointExtractor, same names and defaults as its
 * setters -- slam_lib/include/LidarSlam/SpinningSensorKeypointExtractor.h:44-70, 125-157 */
typedef struct lsa_extract_params_t
{
  int32_t neighbor_width;            /* NeighborWidth = 4 */
  float min_distance_to_sensor;      /* MinDistanceToSensor = 1.5 m */
  float min_beam_surface_angle;      /* MinBeamSurfaceAngle = 10 deg */
  float plane_sin_angle_threshold;   /* PlaneSinAngleThreshold = 0.5 */
  float edge_sin_angle_threshold;    /* EdgeSinAngleThreshold = 0.86 */
  float dist_to_line_threshold;      /* DistToLineThreshold = 0.20 m */
  float edge_depth_gap_threshold;    /* EdgeDepthGapThreshold = 0.15 m */
  float edge_saliency_threshold;     /* EdgeSaliencyThreshold = 1.5 m */
  float edge_intensity_gap_threshold;/* EdgeIntensityGapThreshold = 50 */
} lsa_extract_params_t;

/* KeypointsMatcher::Parameters -- slam_lib/include/LidarSlam/KeypointsMatcher.h:43-77 */
typedef struct lsa_match_params_t
{
  int32_t single_edge_per_ring;      /* SingleEdgePerRing */
  int32_t edge_nb_neighbors;         /* EdgeNbNeighbors */
  int32_t edge_min_nb_neighbors;     /* EdgeMinNbNeighbors */
  int32_t plane_nb_neighbors;        /* PlaneNbNeighbors */
  int32_t blob_nb_neighbors;         /* BlobNbNeighbors */
  int32_t reserved;
  double max_neighbors_distance;     /* MaxNeighborsDistance */
  double edge_max_model_error;       /* EdgeMaxModelError */
  double planarity_threshold;        /* PlanarityThreshold */
  double plane_max_model_error;      /* PlaneMaxModelError */
  double saturation_distance;        /* SaturationDistance (Tukey scale) */
} lsa_match_params_t;

/* Which device-resident keypoint set an operation reads */
#define LSA_SET_RAW_CURRENT 0   /* Slam::CurrentRawKeypoints           (Slam.h:516) */
#define LSA_SET_RAW_PREVIOUS 1  /* Slam::PreviousRawKeypoints          (Slam.h:517) */
#define LSA_SET_WORKING 2       /* Slam::CurrentUndistortedKeypoints   (Slam.h:520) */

/* ------------------------------------------------------------------------- */
/* Context: one per Slam instance, bound to one HIP device and one stream.    */
typedef struct lsa_ctx lsa_ctx;

int lsa_device_count(void);
int lsa_ctx_create(int device_id, lsa_ctx** out);
/* Restricts the calling thread (and the threads it creates afterwards: call it before lsa_ctx_create /
 * lsa_slam_create) to the CPUs of the NUMA node the device is attached to.  A frame is tens of short host <-> device
 * round trips (mailbox polls, pinned staging buffers); from the other socket of a two-socket host they take about
 * twice as long.  Returns the node (>= 0), LSA_E_STATE when the host has no NUMA information or none of the node's
 * CPUs may be used, LSA_E_NO_DEVICE for a bad device.  Optional: nothing else depends on it. */
int lsa_bind_host_to_device(int device_id);
void lsa_ctx_destroy(lsa_ctx* ctx);
const char* lsa_last_error(const lsa_ctx* ctx);
/* Blocks until everything queued on the context's stream has finished. */
int lsa_sync(lsa_ctx* ctx);

/* ------------------------------------------------------------------------- */
/* Seam 1: SpinningSensorKeypointExtractor::ComputeKeyPoints(pc)
 * -- slam_lib/src/SpinningSensorKeypointExtractor.cxx:118-136, called from
 *    Slam::ExtractKeypoints, slam_lib/src/Slam.cxx:784.                       */

/* Copies one scan (firing order, rings interleaved) to the device and makes it
 * the current frame.  On the first usable frame the azimuthal resolution is
 * estimated on the host exactly as EstimateAzimuthalResolution does
 * (SSKE.cxx:593-637) and then frozen in the context (SSKE.cxx:169-170). */
int lsa_upload_frame(lsa_ctx* ctx, const lsa_point_t* pts, int n);

/* The sensor driver's record layout (a sensor_msgs/PointCloud2 of velodyne_pcl::PointXYZIRT, or any record
 * with float x, y, z, intensity, time and a uint16 ring): byte offsets inside a record of point_step bytes. */
typedef struct lsa_wire_layout_t
{
  int32_t point_step;
  int32_t off_x, off_y, off_z, off_intensity, off_ring, off_time;
} lsa_wire_layout_t;

/* lsa_upload_frame for the driver's records, converted on the device as VelodyneToLidarNode::Callback does on
 * the host (ros_wrapping/lidar_conversions/src/VelodyneToLidarNode.cxx:52-112): laser_id = mapping[ring] (or
 * ring when mapping_len == 0), device_id, time = the record's time offset.  When the time field is not usable
 * (last - first <= 1e-8, :74) the time is built from the azimuth advancement exactly as the node does
 * (SpinningFrameAdvancementEstimator, lidar_conversions/src/Utilities.h:62-114, rpm and timestamp_first_packet
 * as its parameters): per ring that is "the first descent of the advancement and everything after it", found on the
 * ring-bucketed frame on the device (portable arc tangent: the time agrees with the node's to rounding).  The first
 * frame (azimuthal resolution estimate) is converted on the host. */
int lsa_upload_wire_frame(lsa_ctx* ctx, const void* records, int n, const lsa_wire_layout_t* layout, const uint16_t* laser_id_mapping,
                          int mapping_len, int device_id, double rpm, int timestamp_first_packet);

/* lsa_upload_frame for the RoboSense driver's organized cloud (height = lasers, width = points per laser, row after row,
 * records of pcl::PointXYZI: float x, y, z, intensity at the given byte offsets -- off_ring and off_time of the layout are
 * not used), converted on the device as RobosenseToLidarNode::Callback does on the host
 * (ros_wrapping/lidar_conversions/src/RobosenseToLidarNode.cxx:58-125): records with a NaN or infinite coordinate are
 * dropped (:82-83); so is a record whose three coordinates equal those of the last point kept (second return of the dual
 * return mode, :87-88); laser_id = mapping[i / width], or RS16's own mapping when there are 16 lasers and no mapping is
 * given, or i / width (:106-109); time = ((i mod (size / height)) / (size / height) - 1) / rpm * 60 (:118-119); the points
 * kept stay in their order.  *n_valid = points kept (= the size of the current frame; 0 and no frame when none is). */
int lsa_upload_robosense_frame(lsa_ctx* ctx, const void* records, int width, int height, const lsa_wire_layout_t* layout, const uint16_t* laser_id_mapping,
                               int mapping_len, int device_id, double rpm, int* n_valid);

/* lsa_upload_frame for a LidarView / ParaView frame, converted on the device as vtkSlam::PolyDataToPointCloud does on the
 * host (paraview_wrapping/Plugin/vtkLidarSlam/vtkSlam.cxx:668-707), from the vtkPolyData's own arrays -- structure
 * of arrays, no LidarPoint cloud is built on the host: xyz = 3 n interleaved coordinates (vtkPoints, float or double),
 * time / laser_id / intensity = the point-data arrays named by the filter (any of the scalar types below).  The
 * frame's end is the largest time; *stamp_us = end * (time_to_seconds * 1e6), a point's time = (t - end) *
 * time_to_seconds, laser_id = mapping[id] (or id when mapping_len == 0); points whose three coordinates are all zero
 * are dropped, the others keep their order.  *n_valid = points kept (= the size of the current frame).  Returns 1
 * when every point was kept ("allPointsAreValid"), 0 when some were dropped, < 0 on error. */
enum { LSA_SCALAR_F32 = 0, LSA_SCALAR_F64 = 1, LSA_SCALAR_U8 = 2, LSA_SCALAR_U16 = 3, LSA_SCALAR_U32 = 4, LSA_SCALAR_I32 = 5 };
int lsa_upload_polydata_frame(lsa_ctx* ctx, int n, const void* xyz, int xyz_type, const void* time, int time_type, const void* laser_id, int laser_type,
                              const void* intensity, int intensity_type, const uint16_t* laser_id_mapping, int mapping_len, double time_to_seconds,
                              uint64_t* stamp_us, int* n_valid);

/* Frame store: keeps scans resident in HBM so that a replay (bench.py) can
 * time the path without the PCIe copy.  Slots are created on demand. */
int lsa_frame_store_put(lsa_ctx* ctx, int slot, const lsa_point_t* pts, int n);
int lsa_frame_store_use(lsa_ctx* ctx, int slot);

/* Number of points of the current frame. */
int lsa_frame_size(const lsa_ctx* ctx);

float lsa_get_azimuthal_resolution(const lsa_ctx* ctx);
void lsa_set_azimuthal_resolution(lsa_ctx* ctx, float rad);

/* Runs a3-a8 of SURVEY.md 8a on the current frame.  The previous call's
 * keypoints become LSA_SET_RAW_PREVIOUS (Slam.cxx:751).  counts[k] = number of
 * keypoints of type k; keypoints stay on the device, ordered ring-major /
 * index-ascending as SSKE.cxx:575-589 pushes them. */
int lsa_extract_keypoints(lsa_ctx* ctx, const lsa_extract_params_t* params, int counts[3]);
/* A further frame of the same Slam::AddFrames call (another LiDAR device, Slam.cxx:753-801): the frame in the context
 * is extracted with `params` (and the azimuthal resolution set for that device) and its keypoints are appended to
 * the RAW_CURRENT sets instead of replacing them -- AggregateFrames(keypoints, false) (Slam.cxx:1512-1578):
 * time += time_offset, then the rigid transform sensor -> BASE (NULL = identity, coordinates untouched).
 * counts[] = keypoints of this frame per type. */
int lsa_extract_keypoints_more(lsa_ctx* ctx, const lsa_extract_params_t* params, const double base_to_lidar[16], double time_offset, int counts[3]);
/* A frame handed over AHEAD of the AddFrame call that will use it (offline / batch replay: the caller holds the
 * next cloud while the current one is being registered).  lsa_upload_frame_begin returns at once: a thread of the
 * context copies the (pageable) cloud into pinned staging memory and enqueues the DMA on a copy stream of its own,
 * into one of three device buffers, so that the upload of frame f + 1 runs beside the registration of frame f.
 * Up to two clouds may be announced before the first of them is used (the replay announces frame f + 1 just before it
 * adds frame f, which it announced one call earlier); a third announcement gives the oldest up.  `pts` must stay valid
 * and unchanged until the frame has been adopted or given up.
 * lsa_upload_frame_ready: 1 once the DMA of the OLDEST announced cloud has been enqueued (lsa_extract_prefetch_uploaded
 * may follow).  lsa_upload_frame_adopt(pts, n): makes an announced cloud the current frame (as lsa_upload_frame would)
 * when it is the very buffer that was announced -- returns 1; clouds announced before it are given up --, returns 0
 * when it was not announced (the caller then calls lsa_upload_frame).  lsa_extract_prefetch_uploaded:
 * lsa_extract_prefetch for the oldest announced cloud. */
int lsa_upload_frame_begin(lsa_ctx* ctx, const lsa_point_t* pts, int n);
int lsa_upload_frame_ready(const lsa_ctx* ctx);
int lsa_upload_frame_adopt(lsa_ctx* ctx, const lsa_point_t* pts, int n);
/* Page-locks a host buffer the caller hands clouds over in again and again (a driver's ring of scan buffers): uploads
 * from it (lsa_upload_frame, lsa_slam_add_frame) are then one DMA at the bus rate, asynchronous to the caller, instead of
 * a staged copy through the runtime's own pinned chunks (VLS-128, 8.2 MB: 0.15 ms against 0.3 ms).  Process-wide
 * (hipHostRegister); lsa_unpin_host_memory before the buffer is freed. */
int lsa_pin_host_memory(void* ptr, size_t bytes);
int lsa_unpin_host_memory(void* ptr);
/* Gives up every cloud announced with lsa_upload_frame_begin and not adopted yet (Slam::Reset: nothing announced before a
 * reset is taken over after it).  A cloud whose buffer was rewritten since it was announced is never adopted either: a
 * sample of its contents is compared (lsa_upload_frame_adopt then returns 0 and the caller uploads). */
int lsa_upload_frame_forget(lsa_ctx* ctx);
/* Frees the buffers the context has outgrown since the last call.  Growth never frees on the spot: hipFree waits for the
 * whole device, also for an ICP iteration that waits on the device for this very process (lsa_icp_gate).  Call it where
 * nothing of the context waits on the device for the host -- the pipeline does at the start of every AddFrame. */
int lsa_collect_garbage(lsa_ctx* ctx);
int lsa_uploads_adopted(const lsa_ctx* ctx);
int lsa_extract_prefetch_uploaded(lsa_ctx* ctx, const lsa_extract_params_t* params);
/* Look-ahead for replay from the frame store: extracts the keypoints of the frame in `slot` on a stream of its own,
 * beside whatever runs on the context's stream (the registration of the current frame), into spare buffers.  The
 * next lsa_extract_keypoints adopts them -- no kernel, no wait -- if it is called for that very frame
 * (lsa_frame_store_use(slot)) with the same parameters, keypoint types and azimuthal resolution; otherwise the
 * look-ahead is dropped and the extraction runs as usual.  Same keypoints either way (the extraction does not depend
 * on the pose).  Call it after the current frame's lsa_extract_keypoints; until the adoption lsa_download_debug is
 * refused (the per-point arrays are being rewritten). */
int lsa_extract_prefetch(lsa_ctx* ctx, int slot, const lsa_extract_params_t* params);
/* how many look-aheads lsa_extract_keypoints has adopted so far on this context */
int lsa_extract_prefetch_adopted(const lsa_ctx* ctx);
/* Keypoint types lsa_extract_keypoints keeps (bit k = type k; Slam::UseKeypoints, Slam.h:406): the others
 * come out empty, exactly as Slam::ExtractKeypoints drops them (Slam.cxx:789-793).  Default: all three. */
int lsa_set_keypoint_types(lsa_ctx* ctx, unsigned type_mask);

/* Copies a device keypoint set to the host.  Returns the number of points
 * written (<= capacity) or a negative error. */
int lsa_download_keypoints(lsa_ctx* ctx, int set, int type, lsa_point_t* out, int capacity);
int lsa_keypoint_count(const lsa_ctx* ctx, int set, int type);

/* SpinningSensorKeypointExtractor::GetDebugArray() (SSKE.cxx:640-680), one
 * array per call, in scan order.  array_id: 0 sin_angle, 1 saliency,
 * 2 depth_gap, 3 intensity_gap, 4/5/6 edge/plane/blob_keypoint,
 * 7/8/9 edge/plane/blob_validity. */
int lsa_download_debug(lsa_ctx* ctx, int array_id, float* out, int capacity);
int lsa_nb_laser_rings(const lsa_ctx* ctx);

/* Rigidly transforms a raw-current keypoint set in place (LIDAR -> BASE,
 * Slam::AggregateFrames(..., false), Slam.cxx:1551-1573) and adds time_offset
 * to every point's time. T is a row-major 4x4. */
int lsa_transform_keypoints(lsa_ctx* ctx, int set, int type, const double T[16], double time_offset);

/* ------------------------------------------------------------------------- */
/* Seam 2: KeypointsMatcher::BuildMatchResiduals(currPoints, kdtree, type)
 * -- slam_lib/src/KeypointsMatcher.cxx:33-74; the kd-tree build it replaces is
 *    KDTreePCLAdaptor::Reset, slam_lib/include/LidarSlam/KDTreePCLAdaptor.h:57-65. */

/* The reference keeps two independent families of kd-trees alive at the same time: the sub-map
 * trees of the rolling grids (rebuilt only on keyframes, RollingGrid.cxx:441) and the trees on the
 * previous scan's keypoints that ComputeEgoMotion builds every frame (Slam.cxx:845-860).  A target
 * therefore lives in a slot: */
#define LSA_TARGET_MAP 0       /* RollingGrid::GetSubMapKdTree() */
#define LSA_TARGET_PREVIOUS 1  /* kdtreePrevious of Slam::ComputeEgoMotion */

/* Sets the kNN target of one keypoint type from host points (the sub-map the
 * host-side RollingGrid produced, Slam.cxx:1003-1037) and builds the device
 * search grid.  Point order is kept: indices seen by the PCA are these. */
int lsa_set_target(lsa_ctx* ctx, int slot, int type, const lsa_point_t* pts, int m);
/* Same, from a device-resident keypoint set (ego-motion registers on the
 * previous frame's raw keypoints, Slam.cxx:845-860): no PCIe traffic. */
/* lsa_set_target without the wait: the caller fills a pinned host buffer owned by the context
 * (lsa_target_staging returns it, grown to `capacity` points; it may be called and filled from another host
 * thread, e.g. the one that extracts the sub-map) and lsa_set_target_staged enqueues the copy of its first m
 * points.  The buffer must stay untouched until the next match of that target has been waited for. */
lsa_point_t* lsa_target_staging(lsa_ctx* ctx, int slot, int type, int capacity);
int lsa_set_target_staged(lsa_ctx* ctx, int slot, int type, int m);
/* Ahead of its use: the staging buffer of (LSA_TARGET_MAP, type) holds m points that will probably become the target
 * (a sub-map extracted for the predicted pose).  They are uploaded and their search grid is built on the look-ahead
 * stream into a spare target; lsa_set_target_staged with the same m and cell size then only swaps it in.  Before the
 * staging buffer is rewritten instead (the prediction did not hold), lsa_drop_target_ahead waits for the copy out of
 * it.  lsa_staged_targets_adopted counts the take-overs. */
int lsa_stage_target_ahead(lsa_ctx* ctx, int slot, int type, int m);
int lsa_drop_target_ahead(lsa_ctx* ctx, int slot, int type);
int lsa_staged_targets_adopted(const lsa_ctx* ctx);
int lsa_set_target_from_set(lsa_ctx* ctx, int slot, int type, int set);
/* Builds ahead of time the targets the NEXT frame's ego-motion will search: the current raw keypoints of the types
 * in type_mask (which the next lsa_extract_keypoints turns into the previous ones) are copied and their search
 * grids built on the look-ahead stream, beside the current frame's registration.  The next
 * lsa_set_target_from_set(LSA_TARGET_PREVIOUS, type, LSA_SET_RAW_PREVIOUS) takes them over if the set was not
 * written since and the cell size is the one they were built with; otherwise it builds as usual.  Same neighbours
 * either way.  lsa_prepared_targets_adopted counts the take-overs. */
int lsa_prepare_previous_targets(lsa_ctx* ctx, unsigned type_mask);
int lsa_prepared_targets_adopted(const lsa_ctx* ctx);
int lsa_target_size(const lsa_ctx* ctx, int slot, int type);
/* The target's points as they were given (Slam::GetTargetSubMap, Slam.h:168): returns the number written. */
int lsa_download_target(lsa_ctx* ctx, int slot, int type, lsa_point_t* out, int capacity);
/* Edge length [m] of the search-grid cells used by the next lsa_set_target* of this type
 * (default 1.0; it is enlarged automatically when the grid would exceed 2^21 cells). */
int lsa_set_target_cell_size(lsa_ctx* ctx, int slot, int type, float cell);
/* Tuning only (results do not depend on it): lanes of a wavefront that cooperate on one query of `type`
 * in the first kNN kernel: 8, 16 or 32 (sparse targets such as edges: more lanes). */
int lsa_set_knn_lanes(lsa_ctx* ctx, int type, int lanes);
/* ... and the number of rounds of that kernel (2: blocks of 3^3 and 5^3 cells; 3: also 7^3) before a query is
 * handed to the second kernel. */
int lsa_set_knn_rounds(lsa_ctx* ctx, int type, int rounds);

/* MatchingResults::NbMatches() and the rejection histogram of an earlier match
 * (KeypointsMatcher.h:100-122) without reading anything back on the frame's
 * critical path: lsa_match_serial names the last match enqueued for a type, and
 * lsa_match_histogram reads that match's histogram later (it waits for the stream).
 * The device keeps the last 16 matches of every type. */
long long lsa_match_serial(const lsa_ctx* ctx, int type);
int lsa_match_histogram(lsa_ctx* ctx, int type, long long serial, int histogram[LSA_MATCH_NSTATUS]);
/* lsa_undistort(H0, H1, t0, t1) followed by lsa_match_types on LSA_SET_WORKING -- what comes between two iterations of
 * the localization ICP (Slam.cxx:1140-1147, 1074-1091) -- with the undistortion inside the search kernel when that kernel
 * reaches every keypoint of the working set (one-launch form, every type that has keypoints asked for, with a target and
 * valid parameters); as the two calls otherwise.  Same keypoints, same matches either way. */
int lsa_match_types_undistorted(lsa_ctx* ctx, int slot, unsigned type_mask, const lsa_match_params_t* p, const double pose[16], int* histograms,
                                const double H0[16], const double H1[16], double t0, double t1);
/* lsa_match_types as ONE launch for all keypoint types, search and model fit in the same kernel (on = 1, the default),
 * as two launches (on = 2: the searches of all types, then their model fits), or (on = 0) as the staged kernels, types
 * side by side on streams.  Same results either way. */
int lsa_set_fused_match(lsa_ctx* ctx, int on);
/* Diagnostics: queries of the last lsa_match that the first kNN kernel handed to the second stage. */
int lsa_match_slow_queries(lsa_ctx* ctx);
/* Diagnostics of the last fused match of `type` (LSA_ROUTE_STATS=1 when the context is created): [0] queries handed
 * to the tail kernel, [1] of them done, [2] second scans, [3] first block beyond shell 2, [4] candidates walked,
 * [5] NEIGHBORS_TOO_FAR by the counts, [6] first block = 3^3 finest cells, [7] longest walk of one lane. */
int lsa_match_route_stats(lsa_ctx* ctx, int type, int out[8]);
/* Diagnostics (LSA_ROUTE_STATS=1): {start, search end, model end, XCC | HW_ID} of the hardware blocks of the last
 * fused match, 100 MHz ticks. */
int lsa_match_trace(lsa_ctx* ctx, unsigned long long* out, int blocks);
/* ... and those of them that ended up scanning the whole target. */
int lsa_match_exhaustive_queries(lsa_ctx* ctx);

/* Replaces a device keypoint set by host points (used by tests and by callers
 * that aggregate several LiDAR devices on the host). */
int lsa_set_keypoints(lsa_ctx* ctx, int set, int type, const lsa_point_t* pts, int k);

/* For every keypoint of `type` in `query_set`: world = pose * X, exact kNN in
 * the target, neighbourhood filtering, PCA model fit, validity tests, residual
 * record (A, P, X, weight).  Records stay on the device for lsa_accumulate.
 * pose = PosePrior, row-major 4x4.  histogram[s] = number of keypoints with
 * MatchStatus s (MatchingResults::RejectionsHistogram). */
int lsa_match(lsa_ctx* ctx, int slot, int type, int query_set, const lsa_match_params_t* params, const double pose[16],
              int histogram[LSA_MATCH_NSTATUS]);

/* One ICP iteration's matching step (the `for (auto k : KeypointTypes)` loops of
 * Slam.cxx:895-912 and 1074-1091): every type in type_mask (bit k = type k) is
 * matched as by lsa_match, the types running concurrently on the device.
 * histograms: NULL, or [3][LSA_MATCH_NSTATUS] (rows of types outside the mask
 * are zero).  With NULL the call only enqueues work and returns without waiting
 * for the device; lsa_accumulate's n_valid then reports the number of matches. */
int lsa_match_types(lsa_ctx* ctx, int slot, unsigned type_mask, int query_set, const lsa_match_params_t* params, const double pose[16],
                    int* histograms);

/* Confidence::LCPEstimator (slam_lib/src/ConfidenceEstimators.cxx:27-65) as Slam::EstimateOverlap calls it
 * (Slam.cxx:1370-1388): every (1 / sampling_ratio)-th point of the current frame is registered into the
 * world (H0, or the H0 -> H1 interpolation of lsa_transform_frame when interpolate != 0), its nearest
 * neighbour is searched in the map-slot target of every type in type_mask that holds points, and the best
 * Gaussian score exp(-d^2 / (2 (leaf / 3)^2)) is averaged.  *overlap = -1 when nothing can be estimated.
 * The reference sums in float under an OpenMP reduction, i.e. in no defined order; agreement is to rounding. */
int lsa_overlap(lsa_ctx* ctx, unsigned type_mask, int interpolate, const double H0[16], const double H1[16], double t0, double t1, float sampling_ratio,
                const double leaf_size[3], float* overlap);

/* MatchingResults::Rejections / Weights of the last lsa_match of `type`
 * (exported by Slam::GetDebugArray, Slam.cxx:635-657).  records (optional,
 * may be NULL) receives 16 doubles per keypoint: A[9] row-major, P[3], X[3],
 * weight; rows of unmatched keypoints are zero. */
int lsa_download_match(lsa_ctx* ctx, int type, uint8_t* status, double* weights, double* records, int capacity);

/* ------------------------------------------------------------------------- */
/* Seam 3: LocalOptimizer::Solve() -- slam_lib/src/LocalOptimizer.cxx:74-102.
 * The device evaluates what Ceres evaluates per LM step for the residual
 * blocks of the last lsa_match calls of the types in type_mask (bit k = type
 * k), in the order EDGE, PLANE, BLOB (Slam.cxx:936-937, 1120-1121):
 *   r_i = A_i (R(rpy) X_i + t - P_i),  rho = weight_i * Tukey(|r_i|^2)
 *   cost = 1/2 sum rho,  g = sum rho' J^T r,  H = sum rho' J^T J (upper part
 *   mirrored, row-major 6x6), w = (x, y, z, rx, ry, rz).
 * The 6-dof trust-region control flow itself stays on the host (it is a
 * 6x6 solve); see lidarslam_amd/csrc/host/lsa_lm.cpp. */
int lsa_accumulate(lsa_ctx* ctx, unsigned type_mask, const double w[6], int want_jacobian, double* cost, double g[6],
                   double H[36], int* n_valid);
/* 1 when the partial sums of lsa_accumulate arrive through coherent host memory the kernel writes directly (the
 * normal case), 0 when that memory could not be mapped and every evaluation falls back to a copy + synchronise
 * (same results, about twice the time per evaluation). */
int lsa_mailbox_active(const lsa_ctx* ctx);

/* LocalOptimizer::SetPosePrior + Solve + GetOptimizedPose (LocalOptimizer.cxx:44-48, 74-109) on the
 * device-resident residual blocks of type_mask: the Ceres trust-region Levenberg-Marquardt loop
 * (DENSE_QR, max_num_iterations = lm_max_iter, TwoDMode holds Z, rX, rY) with every evaluation on
 * the GPU.  summary[0] = num_successful_steps (counts iteration 0, as ceres::Solver::Summary does:
 * == 1 means no step was accepted, Slam.cxx:950), [1] unsuccessful steps, [2] iterations,
 * [3] evaluations; costs[0] initial, [1] final. */
int lsa_solve(lsa_ctx* ctx, unsigned type_mask, const double prior[16], int lm_max_iter, int two_d_mode, double optimized[16],
              int summary[4], double costs[2]);
/* The same solve as ONE launch: the trust-region loop itself runs on the device (every block evaluates its share
 * of the residual blocks, the blocks exchange their partial sums through tagged 8-byte granules, each block runs the
 * 6x6 algebra), the host only reads the result.  prior / pose are the six parameters (x, y, z, rx, ry, rz) of
 * LocalOptimizer::PoseArray (LocalOptimizer.h:99-101).  min_matches: Slam.cxx:919, 1098 skip the optimisation when
 * fewer keypoints matched (skipped = 1, pose = prior).  cost, g, H: the normal equations at the returned pose (what
 * EstimateRegistrationError needs).  Returns LSA_E_STATE when the device gave up (blocks not co-resident in time):
 * the caller then runs lsa_solve / the host-driven loop; nothing was changed. */
typedef struct lsa_solve_result
{
  double pose[6];
  double initial_cost, final_cost;
  double cost, g[6], H[36];
  int num_successful_steps;   /* counts iteration 0, as ceres::Solver::Summary does */
  int num_unsuccessful_steps;
  int num_iterations;
  int num_evaluations;
  int num_matches;            /* residual blocks (successful matches) the problem was built from */
  int skipped;
  int termination;            /* 0 none, 1 not enough matches, 2 gradient tolerance at iteration 0, 3 max iterations,
                                 4 gradient tolerance, 5 min trust region radius, 6 too many invalid steps,
                                 7 parameter tolerance, 8 function tolerance */
  const char* message;
} lsa_solve_result_t;
int lsa_solve_device(lsa_ctx* ctx, unsigned type_mask, const double prior[6], int two_d_mode, int lm_max_iter, int min_matches,
                     lsa_solve_result_t* out);
/* lsa_solve_device in two halves: _begin enqueues the solve (prior == NULL: the start point is the one the gate in
 * front of it will hand over, see lsa_icp_gate), _end waits for the result of the oldest solve begun and not ended.
 * _drop forgets the solve begun last without waiting (its gate was called off: it will never run). */
int lsa_solve_device_begin(lsa_ctx* ctx, unsigned type_mask, const double prior[6], int two_d_mode, int lm_max_iter, int min_matches);
int lsa_solve_device_end(lsa_ctx* ctx, lsa_solve_result_t* out);
int lsa_solve_device_drop(lsa_ctx* ctx);

/* ICP iterations enqueued AHEAD of their inputs (the `for icpIter` loops of Slam::ComputeEgoMotion / Localization,
 * slam_lib/src/Slam.cxx:892-950, 1071-1145: iteration i + 1 needs the pose iteration i ends with).
 *   ticket = lsa_icp_gate(ctx)              a gate on the context's stream; the launches enqueued next with
 *   lsa_match_types_gated(...)              gated inputs -- this match and lsa_solve_device_begin(prior = NULL) -- wait
 *   lsa_solve_device_begin(ctx, .., NULL)   behind it ON THE DEVICE
 *   lsa_icp_post(ctx, ticket, pose, prior, H0, H1, t0, t1)   hands over the pose to search under, the optimiser's start
 *                                           point and (H0 / H1 not NULL) the undistortion the match starts with: they run
 *   lsa_icp_cancel(ctx, ticket)             or calls them off: they do nothing, nobody waits for them
 * so that the kernel-launch path is not between the end of a solve and the next search.  A gate waits 50 ms at most; after
 * that the launches behind it do nothing and lsa_solve_device_end reports LSA_E_GATE.  At most 8 gates in flight.
 * lsa_icp_abandon calls off every gate still waiting and forgets the solves begun behind them (error paths, Reset).
 * lsa_match_types_gated returns 1 (and enqueues nothing) when this match cannot wait behind a gate: the two-launch or
 * staged forms, an empty target, an undistortion that would not reach every keypoint. */
int lsa_icp_gate(lsa_ctx* ctx);
int lsa_icp_post(lsa_ctx* ctx, int ticket, const double pose[16], const double prior[6], const double H0[16], const double H1[16], double t0, double t1);
int lsa_icp_cancel(lsa_ctx* ctx, int ticket);
int lsa_icp_abandon(lsa_ctx* ctx);
int lsa_match_types_gated(lsa_ctx* ctx, int slot, unsigned type_mask, int query_set, const lsa_match_params_t* p, int undistort);

/* The same loops WITHOUT the host between two iterations: a LINK is a gate's block on the device that the solve in front
 * of it fills in itself -- whether the next iteration runs (Slam.cxx:919-923, 950 / 1098-1107, 1151: the solve was not
 * skipped and made a step), the pose from the solve's parameters (Utils::XYZRPYtoIsometry), the next start point
 * (LocalOptimizer::SetPosePrior's IsometryToXYZRPY of that pose) and, localization, Slam::RefineUndistortion under the new
 * pose (Slam.cxx:1322-1352) -- with the arithmetic the host uses on the same result when it arrives (lsa_posemath.h over
 * lsa_pmath.h: the same bits).  A whole loop is enqueued at once:
 *   lsa_match_types(...)                                       iteration 0, pose from the host
 *   t1 = lsa_icp_link(ctx)                                     reserves a block (at most 7 at a time)
 *   lsa_solve_device_begin_linked(.., prior, .., t1, &link)    solve 0 leaves block t1; what is enqueued next waits behind it
 *   lsa_match_types_gated(...)                                 iteration 1: searches under the pose in t1, or does nothing
 *   t2 = lsa_icp_link(ctx); lsa_solve_device_begin_linked(.., NULL, .., t2, &link); ...
 *   lsa_solve_device_begin_linked(.., NULL, .., -1, NULL)      the last one leaves nothing
 * then lsa_solve_device_end once per iteration, in order; the host takes the same decision from every result and, where
 * the device stopped, forgets the rest with lsa_icp_abandon.  Between two iterations there is one kernel boundary. */
typedef struct lsa_icp_link
{
  int refine_undistortion;         /* localization: Slam::RefineUndistortion between two iterations (Undistortion = REFINED) */
  int first;                       /* first solve of the loop: the motion within the frame is `motion` (later solves go on
                                      from what the solve before them left on the device) */
  int have_log;                    /* Slam::InterpolateScanPose (Slam.cxx:1271-1285): LogTrajectory is not empty, */
  double prev_time, cur_time;      /*   its last time and the current frame's [s], */
  double max_extrapolation_ratio;  /*   MaxExtrapolationRatio */
  double previous_world[16];       /* PreviousTworld */
  double motion[16];               /* LinearTransformInterpolator: Time0, Time1, Rot0 (w x y z), Rot1, Trans0[3], Trans1[3] */
} lsa_icp_link_t;
int lsa_icp_link(lsa_ctx* ctx);
int lsa_solve_device_begin_linked(lsa_ctx* ctx, unsigned type_mask, const double prior[6], int two_d_mode, int lm_max_iter, int min_matches,
                                  int leave_ticket, const lsa_icp_link_t* link);
/* Test hook: the block of `ticket` as the device holds it now (waits for the context's stream): words[0] = go, then pose
 * R[9] t[3], start point [6], and the undistortion's constants (64 words in all, doubles unless stated in lsa_posemath.h). */
int lsa_icp_link_peek(lsa_ctx* ctx, int ticket, unsigned long long words[64]);
/* ... and the block the HOST's arithmetic gives for a solve that ended at `x` (its result's skipped / successful steps decide
 * whether the next iteration runs): what lsa_icp_link_peek must show, word for word wherever go == 1 (with go == 0 only
 * word 0 counts).  motion_after (may be NULL): the LinearTransformInterpolator state after the refinement. */
int lsa_icp_link_expected(const double x[6], int skipped, int successful_steps, const lsa_icp_link_t* link, unsigned long long words[64], double motion_after[16]);

/* Diagnostics (LSA_ROUTE_STATS=1): 100 MHz ticks block 0 spent evaluating, exchanging, folding, stepping, summed over
 * the solves so far; [4] evaluations, [5] ticks inside the kernel, [6] solves. */
int lsa_solve_device_trace(lsa_ctx* ctx, unsigned long long out[12]);
/* Test hooks for the two bounded waits on the device, so that the callers' fall-backs can be exercised:
 *   "gate_give_up_every" n   every n-th gate (lsa_icp_gate) gives up at once, as if the host had not answered in 50 ms
 *   "lm_give_up_block" b     workgroup b of the NEXT one-launch solve abandons the exchange (the others then wait their
 *                            20 ms and give up too: lsa_solve_device reports LSA_E_STATE); one shot, -1 = none */
int lsa_debug_set(lsa_ctx* ctx, const char* name, int value);
/* Solves the device gave up on so far (diagnostics; 0 on a healthy run). */
int lsa_solve_device_fallbacks(const lsa_ctx* ctx);
/* Host work for the time the next solve runs on the device: `fn(arg)` is called once, on the calling thread, by the
 * next lsa_solve_device after it has enqueued its kernel and before it waits for the result (enqueueing work that does
 * not depend on the solve, e.g. on another stream).  NULL withdraws it. */
int lsa_solve_device_interlude(lsa_ctx* ctx, void (*fn)(void*), void* arg);
/* LocalOptimizer::EstimateRegistrationError (LocalOptimizer.cxx:112-140) at `pose`: covariance
 * (row-major 6x6, DoF order X,Y,Z,rX,rY,rZ), err[0] position error [m], err[1] orientation error [deg]. */
int lsa_registration_error(lsa_ctx* ctx, unsigned type_mask, const double pose[16], int two_d_mode, double cov[36], double err[2]);

/* ------------------------------------------------------------------------- */
/* Seam 4: undistortion / transforms.                                         */

/* CurrentUndistortedKeypoints = CurrentRawKeypoints (Slam.cxx:984). */
int lsa_reset_working_keypoints(lsa_ctx* ctx);

/* Slam::RefineUndistortion's per-point loop (Slam.cxx:1342-1351): every point
 * of every LSA_SET_WORKING type is transformed in place by the pose
 * interpolated at its `time` between H0 (at t0) and H1 (at t1)
 * (LinearTransformInterpolator::operator(), MotionModel.h:115-129).
 * H0/H1 row-major 4x4.  If t0 == t1 or H0 ~ H1 the interpolator is invalid and
 * H0 is applied to every point, as in the reference. */
int lsa_undistort(lsa_ctx* ctx, const double H0[16], const double H1[16], double t0, double t1);

/* What Slam::Localization starts with (Slam.cxx:980-999, 1026-1029) as ONE launch: lsa_reset_working_keypoints, then
 * lsa_undistort(H0, H1, t0, t1) unless H0 is NULL, then -- unless box_pose is NULL -- the bounding boxes of the three
 * types under box_pose as lsa_keypoint_bboxes_begin(LSA_SET_WORKING, box_pose) leaves them on the device for the
 * device maps (lsa_device_grid_build_submap_begin_for_keypoints, lsa_device_grid_submap_ahead_take); they only travel
 * to the host if lsa_keypoint_bboxes_end asks for them.  Same working keypoints and boxes as the separate calls.
 * lsa_arm_localization_boxes prepares the boxes' words for that launch (one tiny launch): called while the device is
 * idle -- between two frames -- it is off the frame's critical path; lsa_localization_begin does it itself otherwise. */
int lsa_localization_begin(lsa_ctx* ctx, const double H0[16], const double H1[16], double t0, double t1, const double box_pose[16]);
int lsa_arm_localization_boxes(lsa_ctx* ctx);

/* min/max of the `time` field over the working keypoints (Slam::InitUndistortion,
 * Slam.cxx:1291-1300) and bounding box of pose * working keypoints of one type
 * (Slam.cxx:1026-1029). */
int lsa_working_time_range(lsa_ctx* ctx, double* tmin, double* tmax);
/* ... of any keypoint set (the raw sets get theirs from the extraction at no cost) */
int lsa_keypoint_time_range(lsa_ctx* ctx, int set, double* tmin, double* tmax);
int lsa_working_bbox(lsa_ctx* ctx, int type, const double pose[16], float mn[3], float mx[3]);
/* The same for the three keypoint types in one pass and one synchronisation:
 * mn / mx = [type][xyz]. */
int lsa_working_bboxes(lsa_ctx* ctx, const double pose[16], float mn[9], float mx[9]);
/* The same on any keypoint set, split in two: _begin only enqueues the reduction and its read-back, _end
 * waits for exactly that work.  Whatever the caller enqueues in between overlaps it. */
int lsa_keypoint_bboxes_begin(lsa_ctx* ctx, int set, const double pose[16]);
/* Same with the pose interpolated at every point's own time between H0 (at t0) and H1 (at t1), like lsa_undistort:
 * the box the keypoints will have once they are undistorted with that motion. */
int lsa_keypoint_bboxes_begin_interp(lsa_ctx* ctx, int set, const double H0[16], const double H1[16], double t0, double t1);
int lsa_keypoint_bboxes_end(lsa_ctx* ctx, float mn[9], float mx[9]);
/* The boxes of a prediction, for lsa_device_grid_submap_ahead_begin only: as lsa_keypoint_bboxes_begin(_interp when H1 is
 * given) but enqueued on the context's look-ahead stream (where the device maps work) and never copied to the host, so that
 * nothing of the speculation sits on the context's own stream.  _mark, called by the thread that drives the context once
 * the set's keypoints are enqueued, names the point from which they exist; lsa_keypoint_boxes_predicted may then be
 * called from any thread. */
int lsa_keypoint_boxes_predicted_mark(lsa_ctx* ctx);
int lsa_keypoint_boxes_predicted(lsa_ctx* ctx, int set, const double H0[16], const double H1[16], double t0, double t1);

/* Slam::TransformPointCloud (Slam.cxx:1491-1509) on a device keypoint set:
 * writes pose * set to `out` on the host. */
int lsa_download_transformed(lsa_ctx* ctx, int set, int type, const double pose[16], lsa_point_t* out, int capacity);
/* The same for the three types of a set at once and without waiting: the device writes the transformed
 * points straight into pinned host buffers owned by the context.  lsa_staged_transformed waits for that
 * work only (it may be called from another host thread, e.g. the one that inserts the points into a map)
 * and returns the buffer of one type; it stays valid until the next lsa_stage_transformed. */
int lsa_stage_transformed(lsa_ctx* ctx, int set, const double pose[16]);
int lsa_staged_transformed(lsa_ctx* ctx, int type, const lsa_point_t** pts, int* n);

/* Slam::AggregateFrames(frames, true) (Slam.cxx:1512-1578): the whole current
 * frame to WORLD.  interpolate != 0: per-point pose between H0 (t0) and H1
 * (t1); else rigid H0. */
int lsa_transform_frame(lsa_ctx* ctx, int interpolate, const double H0[16], const double H1[16], double t0, double t1,
                        lsa_point_t* out, int capacity);
/* Same for a further frame of a multi-device AddFrames call: every point's time is shifted by time_offset first
 * (frame stamp - first frame's stamp, Slam.cxx:1536-1549) and stored shifted. */
int lsa_transform_frame_at(lsa_ctx* ctx, int interpolate, const double H0[16], const double H1[16], double t0, double t1, double time_offset,
                           lsa_point_t* out, int capacity);

/* Device self-test of the arithmetic the bit-exact parity rests on: evaluates fn on the GPU for n
 * inputs.  fn: 0 lsa_sin(x) 1 lsa_cos(x) 2 lsa_atan2(y, x) 3 (float)sqrt((float)x)
 * 4 (float)x / (float)y  5 sqrt(x)  6 x / y.  Results as doubles. */
int lsa_selftest_math(lsa_ctx* ctx, int fn, const double* x, const double* y, int n, double* out);
/* Diagnostic: `blocks` single-wave workgroups sleep-spinning for `ms` (<= 2000) milliseconds on a side stream. */
int lsa_selftest_keep_busy(lsa_ctx* ctx, int ms, int blocks);

/* ------------------------------------------------------------------------- */
/* Per-kernel timing of the last call sequence (HIP events on the context's   */
/* stream), for bench.py's roofline block.                                    */
typedef struct lsa_kernel_stat_t
{
  char name[48];
  uint32_t launches;
  double total_ms;
  double bytes;   /* algorithmic bytes summed over the launches */
} lsa_kernel_stat_t;
int lsa_profile_enable(lsa_ctx* ctx, int on);
/* What a pair of HIP events measures around nothing on the context's stream [us] (calibrated when profiling is switched
 * on; every scope's time is what its events measure minus this). */
double lsa_profile_event_overhead_us(const lsa_ctx* ctx);
/* Events cost a few microseconds per scope on a path that is launch bound: time only the scope `scope`,
 * and only one launch in `every` of it (all launches are counted; total_ms is scaled to all of them). */
int lsa_profile_select(lsa_ctx* ctx, const char* scope, int every);
int lsa_profile_reset(lsa_ctx* ctx);
int lsa_profile_get(lsa_ctx* ctx, lsa_kernel_stat_t* out, int capacity);

/* ------------------------------------------------------------------------- */
/* Pipeline level: LidarSlam::Slam behind a C handle (what a ctypes / cgo /
 * JNI user binds).  Mirrors Slam::AddFrame / GetWorldTransform
 * -- slam_lib/include/LidarSlam/Slam.h:111-146.                              */
typedef struct lsa_slam lsa_slam;

int lsa_slam_create(int device_id, lsa_slam** out);
void lsa_slam_destroy(lsa_slam* s);
const char* lsa_slam_last_error(const lsa_slam* s);
/* Generic parameter access by the reference's setter name without "Set",
 * e.g. "EgoMotion", "Undistortion", "LocalizationICPMaxIter", "NbThreads",
 * "VoxelGridLeafSizeEdges" ...  Returns LSA_E_ARG for an unknown name. */
int lsa_slam_set_param(lsa_slam* s, const char* name, double value);
int lsa_slam_get_param(const lsa_slam* s, const char* name, double* value);
void lsa_slam_reset(lsa_slam* s, int reset_log);
/* Slam::ClearMaps: empties the three keypoint maps; poses and parameters stay. */
int lsa_slam_clear_maps(lsa_slam* s);
/* Slam::AddFrame: frame = host scan, stamp_us = pcl header.stamp, seq = header.seq. */
int lsa_slam_add_frame(lsa_slam* s, const lsa_point_t* pts, int n, uint64_t stamp_us, uint32_t seq);
/* Same on a scan already resident in the frame store of the underlying context. */
int lsa_slam_store_frame(lsa_slam* s, int slot, const lsa_point_t* pts, int n);
int lsa_slam_add_stored_frame(lsa_slam* s, int slot, uint64_t stamp_us, uint32_t seq);
/* Replay from the frame store: names the slot of the frame that will be added after the next one, so that its
 * keypoints are extracted beside the registration of that frame (lsa_extract_prefetch).  Purely a scheduling hint:
 * the results are the same with or without it, and a hint that does not come true costs one wasted extraction. */
int lsa_slam_hint_next_stored_frame(lsa_slam* s, int slot);
/* The same for replay from HOST clouds: announces the cloud of the lsa_slam_add_frame call after the next one.  Its
 * upload starts at once and runs beside the registration of the frame in between (pinned staging, a copy stream and a
 * thread of its own: lsa_upload_frame_begin), its keypoints are extracted as soon as it has arrived; the
 * lsa_slam_add_frame that is handed this very buffer (same pointer, same size) takes both over.  The buffer must stay
 * valid and unchanged until that call returns.  Results are the same with or without the hint. */
int lsa_slam_hint_next_frame(lsa_slam* s, const lsa_point_t* pts, int n);
/* Slam::GetWorldTransform: row-major 4x4 + time [s]. */
int lsa_slam_get_world_transform(const lsa_slam* s, double T[16], double* time);
int lsa_slam_get_covariance(const lsa_slam* s, double cov[36]);
/* Slam::GetKeypoints(k, world): number of points written. */
int lsa_slam_get_keypoints(lsa_slam* s, int type, int world, lsa_point_t* out, int capacity);
/* Slam::GetRegisteredFrame. */
int lsa_slam_get_registered_frame(lsa_slam* s, lsa_point_t* out, int capacity);
/* Slam::GetDebugArray entries "EgoMotion: <type> matches" / "Localization: ...". */
int lsa_slam_get_match_status(lsa_slam* s, int localization, int type, uint8_t* status, double* weights, int capacity);
/* Slam::GetDebugInformation-like counters and stage timings [s]:
 * out[0] total, [1] extract, [2] ego_icp, [3] ego_lm, [4] loc_icp, [5] loc_lm,
 * [6] undistort, [7] submap, [8] maps, [9] ego_iters, [10] loc_iters,
 * [11] lm_evals, [12] total matched, [13] keyframe counter, [14] wait for the
 * previous keyframe's asynchronous map insertion, [15] duration of that insertion. */
int lsa_slam_get_stats(const lsa_slam* s, double out[16]);

/* The remaining result getters of Slam.h:141-189.
 * - GetLatencyCompensatedWorldTransform (Slam.cxx:555-590): the last pose extrapolated by the duration of the last
 *   AddFrame (parameter "Latency").
 * - SetWorldTransformFromGuess (Slam.cxx:490-501).
 * - GetTrajectory / GetCovariances (Slam.cxx:593-605): rows of 17 doubles (row-major 4x4 + time [s]) and of 36;
 *   either pointer may be NULL; returns the number of logged poses (parameter "LoggingTimeout": 0 keeps two).
 * - GetDebugInformation (Slam.cxx:610-633): [0..1] ego-motion edges / planes used, [2..4] localization edges /
 *   planes / blobs used, [5] position error, [6] orientation error, [7] overlap, [8] comply motion limits,
 *   [9] latency [s].
 * - GetMap(k, clean) / GetTargetSubMap(k) (Slam.cxx:670-688): return the full size, write at most capacity points. */
int lsa_slam_get_latency_compensated_world_transform(const lsa_slam* s, double T[16], double* time);
/* Several LiDAR devices on one platform (Slam.h:133-138, 239-250; Slam.cxx:753-801, 1512-1578).
 * - lsa_slam_add_frames = Slam::AddFrames: one frame per device (at most 16), each with its own stamp; the device is
 *   the device_id of the frame's first point; the pose is dated by the first frame's stamp.  Every frame is extracted
 *   with its device's extractor and the keypoints are merged in frame order, moved to BASE by the device's offset and
 *   shifted in time by (its stamp - the first stamp).  A frame of a device without an extractor uses the default one
 *   when no other device has one, and is ignored otherwise, as in the reference.
 * - lsa_slam_set_extractor_param = SetKeyPointsExtractor(extractor, deviceId) + the extractor's setter `name`
 *   ("NeighborWidth" ... "EdgeIntensityGapThreshold", "AzimuthalResolution"); device 0 always has an extractor.
 * - lsa_slam_set_base_to_lidar_offset / get = SetBaseToLidarOffset / GetBaseToLidarOffset(deviceId): rigid transform
 *   from the sensor to BASE (identity for a device nobody configured), device_id in [0, 255]. */
int lsa_slam_add_frames(lsa_slam* s, const lsa_point_t* const* pts, const int* n, const uint64_t* stamp_us, const uint32_t* seq, int nframes);
int lsa_slam_set_extractor_param(lsa_slam* s, int device_id, const char* name, double value);
int lsa_slam_get_extractor_param(const lsa_slam* s, int device_id, const char* name, double* value);
int lsa_slam_set_base_to_lidar_offset(lsa_slam* s, const double T[16], int device_id);
int lsa_slam_get_base_to_lidar_offset(const lsa_slam* s, double T[16], int device_id);
int lsa_slam_set_world_transform_from_guess(lsa_slam* s, const double T[16]);
int lsa_slam_get_trajectory(const lsa_slam* s, double* poses, double* covariances, int capacity);
int lsa_slam_get_debug_information(lsa_slam* s, double out[10]);
int lsa_slam_get_map(lsa_slam* s, int type, int clean, lsa_point_t* out, int capacity);
int lsa_slam_get_target_submap(lsa_slam* s, int type, lsa_point_t* out, int capacity);
lsa_ctx* lsa_slam_context(lsa_slam* s);


/* ------------------------------------------------------------------------- */
/* LidarSlam::RollingGrid ON THE DEVICE (slam_lib/include/LidarSlam/RollingGrid.h:63-212, slam_lib/src/RollingGrid.cxx): the
 * rolling voxel map of one keypoint type as ONE array of voxels sorted by (outer voxel index, leaf voxel index), living
 * in the memory of the context it was created on.  Add / Roll / ClearOldPoints / BuildSubMapKdTree are sequences of
 * kernels on the context's look-ahead stream (runs of the batch sorted in LDS and merged by rank, one thread per voxel
 * folding the batch's points for it in arrival order through the reference's per-point rule, old and new voxels merged
 * by rank, stable compactions), ordered by events against the context's stream wherever the two share data; the sub-map is written straight into a kNN target of the context.
 * One grid is driven by one host thread at a time (not necessarily the context's).  Points come out in KEY ORDER (outer index, then leaf index) where the reference
 * hands them out in its hash containers' iteration order -- a defined order in place of an accidental one, adopted by
 * the oracle and the host grid as well ("OrderedMaps").  Sampling modes FIRST, LAST, MAX_INTENSITY, CENTER_POINT;
 * CENTROID (whose reference loop is quadratic in the batch size, RollingGrid.cxx:282-297) is refused: lsa_slam keeps
 * that mode on the host grid.  Parameters by the reference's setter names: "GridSize", "VoxelResolution", "LeafSize",
 * "MinFramesPerVoxel", "Sampling", "DecayingThreshold". */
typedef struct lsa_device_grid lsa_device_grid;
int lsa_device_grid_create(lsa_ctx* ctx, lsa_device_grid** out);
void lsa_device_grid_destroy(lsa_device_grid* g); /* before lsa_ctx_destroy of its context */
int lsa_device_grid_set(lsa_device_grid* g, const char* name, double value);
double lsa_device_grid_get_param(const lsa_device_grid* g, const char* name);
/* RollingGrid::Reset(position) / Clear(). */
int lsa_device_grid_reset(lsa_device_grid* g, const float position[3]);
int lsa_device_grid_clear(lsa_device_grid* g);
/* RollingGrid::Size() (waits for the modifications enqueued so far). */
int lsa_device_grid_size(lsa_device_grid* g);
/* RollingGrid::Add(pointcloud, fixed, currentTime, roll) from host points ... */
int lsa_device_grid_add(lsa_device_grid* g, const lsa_point_t* pts, int n, int fixed, double time, int roll);
/* ... and from a keypoint set of the context moved by `pose` (Slam::UpdateMapsUsingTworld, Slam.cxx:1178-1222): nothing
 * leaves the device and nothing is waited for. */
int lsa_device_grid_add_keypoints(lsa_device_grid* g, int set, int type, const double pose[16], double time);
/* The same in two steps, for a caller that hands the insertion proper to another host thread: _stage_keypoints (on the
 * context's thread) reads the keypoints -- the set may be rewritten right after --, _add_staged inserts them. */
int lsa_device_grid_stage_keypoints(lsa_device_grid* g, int set, int type, const double pose[16]);
int lsa_device_grid_add_staged(lsa_device_grid* g, double time);
/* _stage_keypoints for the keypoint types of a keyframe together (grids[i] takes type types[i] of the set): one launch */
int lsa_device_grid_stage_keypoints_all(lsa_device_grid* const* grids, const int* types, int count, int set, const double pose[16]);
/* ... for up to three maps of one context at once: one launch of every step serves all of them (a block row per map). */
int lsa_device_grid_add_staged_all(lsa_device_grid* const* grids, int count, double time);
int lsa_device_grid_roll(lsa_device_grid* g, const float min_point[3], const float max_point[3]);
int lsa_device_grid_clear_old_points(lsa_device_grid* g, double current_time);
/* RollingGrid::Get(clean): points written. */
int lsa_device_grid_get(lsa_device_grid* g, int clean, lsa_point_t* out, int capacity);
/* RollingGrid::BuildSubMapKdTree(): the sub-map (the whole map when min_point == NULL) becomes the kNN target (slot,
 * type) of the context; returns its size.  lsa_device_grid_submap_valid: RollingGrid::IsSubMapKdTreeValid(). */
int lsa_device_grid_build_submap(lsa_device_grid* g, const float min_point[3], const float max_point[3], int min_nb_points, int slot, int type);
/* The same in two steps: _begin enqueues the extraction, _end waits for it and returns the size -- several grids build
 * side by side.  _begin_for_keypoints takes the box of keypoint type `box_type` as lsa_keypoint_bboxes_begin (which must
 * come right before, and needs no lsa_keypoint_bboxes_end then) left it on the device: nothing is read back. */
int lsa_device_grid_build_submap_begin(lsa_device_grid* g, const float min_point[3], const float max_point[3], int min_nb_points, int slot, int type);
int lsa_device_grid_build_submap_begin_for_keypoints(lsa_device_grid* g, int box_type, int min_nb_points, int slot, int type);
int lsa_device_grid_build_submap_end(lsa_device_grid* g);
int lsa_device_grid_submap_valid(lsa_device_grid* g);
/* Sub-maps AHEAD of time.  The sub-map only depends on the outer voxels the keypoints' box touches, and the predicted pose
 * gives that box to a voxel long before the localization asks: _ahead_begin (right after lsa_keypoint_bboxes_begin(_interp)
 * under the PREDICTED pose, no lsa_keypoint_bboxes_end) extracts the sub-map for the box of keypoint type `box_type` into
 * the context's spare map target, behind the grid's last insertion; _ahead_poll (non-blocking; returns 0 nothing, 1
 * extraction on its way, 2 search grid enqueued) is called now and then meanwhile (or _ahead_wait from a thread of its own); _ahead_take (right after
 * lsa_keypoint_bboxes_begin under the ACTUAL pose) compares the two voxel ranges on the device: when they are the same the
 * spare target becomes target (slot, type), *taken = 1 and the size is returned; otherwise *taken = 0 and the caller
 * extracts the sub-map as usual.  The same sub-map either way. */
int lsa_device_grid_submap_ahead_begin(lsa_device_grid* g, int box_type, int min_nb_points, int type);
int lsa_device_grid_submap_ahead_poll(lsa_device_grid* g);
/* ... of several maps of one context: when all their sizes have arrived, their search grids are built by one sequence of
 * launches (1 while a size is missing, 2 when done). */
int lsa_device_grid_submap_ahead_poll_all(lsa_device_grid* const* grids, int count);
int lsa_device_grid_submap_ahead_wait(lsa_device_grid* g); /* blocking form of _poll, for a thread with nothing else to do */
int lsa_device_grid_submap_ahead_take(lsa_device_grid* g, int box_type, int min_nb_points, int slot, int type, int* taken);
/* _take in two steps (several maps: every comparison is enqueued before any is waited for): _take_begin returns 1 when a
 * comparison is on its way, 0 when nothing fits; _take_end waits for it. */
int lsa_device_grid_submap_ahead_take_begin(lsa_device_grid* g, int box_type, int min_nb_points, int slot, int type);
int lsa_device_grid_submap_ahead_take_end(lsa_device_grid* g, int* taken);

/* ---- the rolling voxel map (host) ---------------------------
 * LidarSlam::RollingGrid -- slam_lib/include/LidarSlam/RollingGrid.h:63-212,
 * slam_lib/src/RollingGrid.cxx.  Map maintenance runs on host threads beside the
 * device work (once per keyframe); the handle below is the class the pipeline
 * uses, exposed so that a maintainer can swap it in on its own and so that the
 * tests can compare it with the oracle call by call (no GPU involved).
 * Parameter names: "GridSize", "VoxelResolution", "LeafSize",
 * "MinFramesPerVoxel", "Sampling" (0 FIRST .. 4 CENTROID), "DecayingThreshold", and "AddThreads" (host threads
 * Add uses on a big cloud; an implementation knob, the map is the same for any value). */
typedef struct lsa_rolling_grid lsa_rolling_grid;
lsa_rolling_grid* lsa_rolling_grid_create(void);
void lsa_rolling_grid_destroy(lsa_rolling_grid* g);
int lsa_rolling_grid_set(lsa_rolling_grid* g, const char* name, double value);
/* RollingGrid::Reset (position may be NULL), ::Clear, ::Size */
void lsa_rolling_grid_reset(lsa_rolling_grid* g, const float position[3]);
void lsa_rolling_grid_clear(lsa_rolling_grid* g);
int lsa_rolling_grid_size(const lsa_rolling_grid* g);
/* RollingGrid::Roll, ::Add (RollingGrid.cxx:117-318), ::ClearOldPoints */
void lsa_rolling_grid_roll(lsa_rolling_grid* g, const float min_point[3], const float max_point[3]);
int lsa_rolling_grid_add(lsa_rolling_grid* g, const lsa_point_t* pts, int n, int fixed, double current_time, int roll);
void lsa_rolling_grid_clear_old_points(lsa_rolling_grid* g, double current_time);
/* RollingGrid::Get(clean): returns the number of points written */
int lsa_rolling_grid_get(const lsa_rolling_grid* g, int clean, lsa_point_t* out, int capacity);
/* RollingGrid::BuildSubMapKdTree: whole map when min_point is NULL, else the
 * bounding-box overload (RollingGrid.cxx:363-442).  Returns the sub-map size. */
int lsa_rolling_grid_build_submap(lsa_rolling_grid* g, const float min_point[3], const float max_point[3], int min_nb_points);
/* IsSubMapKdTreeValid / GetSubMap */
int lsa_rolling_grid_submap_valid(const lsa_rolling_grid* g);
int lsa_rolling_grid_submap(const lsa_rolling_grid* g, lsa_point_t* out, int capacity);

/* ------------------------------------------------------------------------- */
/* Synthetic spinning-LiDAR sequences; host only, no GPU.      */
/* model: 16 (VLP-16, 16x1800), 64 (HDL-64, 64x2048), 128 (VLS-128, 128x2048) */
int lsa_synth_sensor(int model, int* nrings, int* ncols, double* el_min_deg, double* el_max_deg);
/* Returns the number of points written, -1 unknown model, -2 capacity. */
int lsa_synth_frame(int model, uint64_t seed, int frame, lsa_point_t* out, int capacity, uint64_t* stamp_us);
/* Ground-truth BASE pose at the stamp of `frame` relative to frame 0. */
void lsa_synth_pose(int frame, double T[16]);

#ifdef __cplusplus
}
#endif
#endif /* LIDARSLAM_AMD_H */
