// lsa_ctx.h -- device context of liblidarslam_amd (one per Slam instance).
//
// Data layout in HBM (all owned by the context, sized once per capacity step):
//   frame        AoS lsa_point_t[N]           the scan as handed over (32 B / point)
//   ring-major   float4 xyzi[N]               x,y,z,intensity bucketed by laser ring,
//                uint32 orig[N], uint16 ring_of[N]     arrival order kept inside a ring
//   scores       float angle/depth_gap/saliency/intensity_gap [N], uint8 valid[N], label[N]
//   keypoints    AoS lsa_point_t [3 sets][3 types][N]  raw current / raw previous / working
//   target[k]    AoS points + float4 xyz|laser (gather friendly) + dense search grid
//                (cell_start[ncells+1], cell-sorted float4 xyz|index)
//   match[k]     SoA residual records rec[16][K] (A 9, P 3, X 3, weight), status[K]
//   reduce       per-block partials of the normal equations, 29 doubles each
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/lidarslam_amd.h"

namespace lsa
{

constexpr int kMaxRings = 512;            // laser_id < kMaxRings (largest spinning sensors have 128)
constexpr int kMaxRingPoints = 8192;      // points per ring handled by the in-LDS labelling kernel
constexpr int kBucketChunk = 1024;        // points per block of the ring bucketing kernels
constexpr int kGridLevels = 3;            // search grid resolutions: cell, 4 x cell, 16 x cell
constexpr int kCellCap = 1 << 22;         // max cells of the finest kNN search grid
__host__ __device__ constexpr int grid_level_cells(int level) { return level == 0 ? kCellCap : level == 1 ? (kCellCap >> 6) + 64 : (kCellCap >> 12) + 64; }
constexpr int kKnnMax = 16;               // neighbours per query the kNN buffers hold
constexpr int kAccumBlocks = 96;          // grid of the normal-equation kernel (grid-stride); the host folds the blocks' partial sums
constexpr int kAccumBlocksMax = 256;      // what the buffers hold (the grid size can be tuned with LSA_ACCUM_BLOCKS)
constexpr int kMailboxStride = 64;        // 8-byte granules per block in the mailbox: 2 x 29 used, {tag, half of a double} each
constexpr int kLmBlocksMax = 128;         // grid limit of the one-launch LM solve (k_lm_solve); LSA_LM_BLOCKS tunes the grid
constexpr int kLmBlocks = 64;
constexpr int kLmThreads = 512;           // threads of a workgroup of k_lm_solve (8 wavefronts: two per SIMD)
constexpr int kLmOut = 48;                // doubles the LM kernel hands to the host (96 granules)
constexpr int kLmMailRing = 8;            // result mailboxes: solves enqueued one behind the other (lsa_icp_link) each write their own
constexpr int kHistRing = 256;  // half of it is cleared at a time (two fills on the ICP's stream): once in 128 matches
constexpr int kAccumVals = 29;            // cost, g[6], H upper[21], nvalid
constexpr int kGateRing = 8;              // gates (ICP iterations enqueued ahead of their inputs) a context can have in flight
constexpr int kGateWords = 64;            // 8-byte words of a gate block on the device
constexpr int kGateGranules = 128;        // 8-byte granules {sequence number, half a word} of a gate block in host memory

struct GridDesc
{
  float origin[3];
  float cell;
  float inv_cell;
  int dims[3];
  int ncells;
  int npoints;
};

// One resolution of the search grid.  A spinning-LiDAR cloud spans three orders of magnitude in density
// (centimetres between keypoints next to the sensor, tens of metres at range): level 0 answers the dense
// neighbourhoods, the coarser levels let sparse ones expand over empty space in a few shells.
struct GridLevel
{
  float4* sorted = nullptr;        // cell-sorted x,y,z, index bits
  uint32_t* cell_of = nullptr;     // cell id per point
  uint32_t* cell_start = nullptr;  // max_cells + 1
  uint32_t* cell_fill = nullptr;   // max_cells
  uint32_t* block_sums = nullptr;
  int max_cells = 0;
};

struct Target
{
  lsa_point_t* pts = nullptr;   // AoS as given (order kept)
  float4* xyzl = nullptr;       // x,y,z, laser_id bits
  GridLevel lv[kGridLevels];
  GridDesc* desc = nullptr;     // device, [kGridLevels]
  int* bbox_bits = nullptr;     // 6 ordered ints
  int m = 0;
  int cap = 0;
  float cell_hint = 1.0f;
  bool dirty = false;           // points changed, search grid not rebuilt yet
};

struct MatchBuf
{
  double* rec = nullptr;       // [16][cap]
  uint8_t* status = nullptr;   // [cap]
  int* knn_idx = nullptr;      // [kKnnMax][cap] neighbour indices, ascending (distance, index)
  float* knn_d2 = nullptr;     // [kKnnMax][cap]
  int* knn_cnt = nullptr;      // [cap]
  int* slow_list = nullptr;    // [cap] queries handed to the second kNN kernel ...
  float4* slow_pts = nullptr;  // [cap] ... in target coordinates, w = upper bound of the k-th squared distance
  int k = 0;                   // number of queries of the last match
  int cap = 0;
  double sat = 1.0;
  bool valid = false;
};

// A frame on its way from the caller's (pageable) cloud to the device, ahead of the AddFrame call that will use it:
// the uploader thread copies it into pinned memory and enqueues the DMA on the copy stream, the context's own
// thread never waits for either.  Three of them: the current frame, the one uploaded ahead, and room for the hint
// that may come before the current one has been given up.
struct FrameInbox
{
  lsa_point_t* dev = nullptr;     // device buffer
  lsa_point_t* pinned = nullptr;  // pinned staging buffer the DMA reads
  int cap = 0;
  int n = 0;
  const lsa_point_t* src = nullptr;  // the caller's buffer (identity of the frame; read by the uploader thread only)
  unsigned long long fingerprint = 0;  // of the cloud's contents when it was copied (a sample of its points): a buffer rewritten in place is not adopted
  hipEvent_t ev = nullptr;           // recorded on the copy stream behind the DMA
  std::atomic<int> state{0};         // 0 free, 1 posted to the uploader, 2 DMA enqueued and event recorded, -1 failed
};

struct KernelStat
{
  std::string name;
  uint32_t launches = 0;
  uint32_t timed = 0;     // launches that carried events
  double total_ms = 0;    // of the timed launches
  double bytes = 0;
};

struct PendingEvent
{
  int stat;
  hipEvent_t a, b;
};

}  // namespace lsa

struct lsa_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;

  // frame
  lsa_point_t* frame = nullptr;      // current frame (may alias a store slot)
  lsa_point_t* frame_own = nullptr;  // buffer owned for lsa_upload_frame
  int frame_n = 0;
  int cap_n = 0;  // capacity of all N-sized buffers
  std::vector<std::pair<lsa_point_t*, int>> store;
  std::vector<int> store_cap;  // points each slot's buffer holds
  float az_res = 0.f;
  int nb_rings_seen = 0;

  // ring-major
  float4* xyzi = nullptr;
  uint32_t* orig = nullptr;
  uint16_t* ring_of = nullptr;
  uint32_t* block_hist = nullptr;  // [nblocks][kMaxRings]
  int* ring_start = nullptr;       // [kMaxRings + 1]
  int* ring_len = nullptr;         // [kMaxRings]
  // one 64-byte block read back per extraction: [0..3] keypoint counts, [4..11] ring_meta, [12..15] time range bits
  int* extract_out = nullptr;
  // frames uploaded ahead of their AddFrame (lsa_upload_frame_begin): a pinned, triple-buffered inbox filled by a thread
  // of its own over a copy stream
  lsa::FrameInbox inbox[3];
  std::deque<int> inbox_queue;  // slots uploaded ahead and not adopted yet, oldest first (at most two: the cloud of the
                                // next AddFrame, announced during the previous one, and the one after it)
  int inbox_current = -1;  // slot the current frame lives in (-1: not an inbox frame)
  hipStream_t copy_stream = nullptr;
  std::thread uploader;
  std::mutex up_mutex;
  std::condition_variable up_cv, up_done;
  std::deque<int> up_jobs;
  bool up_quit = false;
  // the uploader's helpers: a cloud is staged and sent in up_parts pieces side by side (a thread each), so that 8 MB are on
  // their way in a quarter of the time one thread's memcpy takes -- the next frame's extraction can only start behind them
  std::vector<std::thread> up_helpers;
  std::condition_variable up_help_cv, up_help_done;
  struct UploadSplit { const char* src = nullptr; char* pinned = nullptr; char* dev = nullptr; size_t points = 0; int parts = 1; unsigned long long seq = 0; int done = 0; bool ok = true; } up_split;
  int up_parts = 4;
  int uploads_adopted = 0;
  // look-ahead extraction (lsa_extract_prefetch): the next frame's keypoints are extracted on a stream of their own
  // while the current frame is registered; lsa_extract_keypoints adopts them when it is called for that very frame
  hipStream_t prefetch_stream = nullptr;
  hipEvent_t ev_prefetch = nullptr;
  lsa_point_t* kp_next[3] = {nullptr, nullptr, nullptr};  // capacity cap_n, like the keypoint sets they are swapped with
  int* extract_out_next = nullptr;  // device, 16 ints
  int* host_next = nullptr;         // pinned, 16 ints
  bool prefetch_pending = false;
  int prefetch_adopted = 0;  // look-aheads that were adopted so far
  const lsa_point_t* prefetch_frame = nullptr;
  int prefetch_n = 0;
  lsa_extract_params_t prefetch_params{};
  float prefetch_az = 0.f;
  unsigned prefetch_mask = 0;
  int* ring_meta = nullptr;        // = extract_out + 4: [0] nrings, [1] max laser id, [2] error flags
  float* score[4] = {nullptr, nullptr, nullptr, nullptr};  // angle, depth_gap, saliency, intensity_gap
  uint8_t* valid = nullptr;
  uint8_t* label = nullptr;
  int* ring_counts = nullptr;  // [kMaxRings][3]
  int* kp_count_dev = nullptr; // = extract_out: [3]
  unsigned kp_type_mask = 7;   // keypoint types the extraction keeps (Slam::UseKeypoints)

  // keypoint sets [set][type]
  lsa_point_t* kp[3][3] = {};
  int kp_n[3][3] = {};
  // [min, max] of the points' time field per set, kept while it is known (lsa_working_time_range)
  double kp_time[3][2] = {};
  bool kp_time_valid[3] = {false, false, false};

  // lsa_target_staging: pinned host buffers a target's points are written into before lsa_set_target_staged
  lsa_point_t* tstage[6] = {};
  int tstage_cap[6] = {};
  lsa::Target target[12];  // [slot * 3 + type]: slot 0 = map sub-maps (localization), slot 1 = previous scan (ego-motion);
                          // [6 + type] = the previous-scan targets of the NEXT frame, built ahead (lsa_prepare_previous_targets)
                          // [9 + type] = sub-map targets uploaded and built ahead of their use (lsa_stage_target_ahead)
  bool map_ahead_ready[3] = {false, false, false};
  int map_ahead_adopted = 0;
  hipEvent_t ev_map_ahead[3] = {nullptr, nullptr, nullptr};
  // content versions of the keypoint sets: every write takes a new number, the shift of the current keypoints to
  // the previous ones carries it along -- that is how a target built ahead knows it still describes the set
  unsigned long long kp_clock = 0;
  unsigned long long kp_ver[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  unsigned long long spare_ver[3] = {0, 0, 0};
  bool spare_ready[3] = {false, false, false};
  int spare_adopted = 0;
  hipEvent_t ev_spare = nullptr, ev_kp_ready = nullptr;
  lsa::MatchBuf match[3];

  int accum_blocks = lsa::kAccumBlocks;
  double* partials = nullptr;  // [kAccumBlocksMax][kAccumVals]
  double* reduce_out = nullptr;
  double* host_pinned = nullptr;  // >= 64 doubles, pinned
  // coherent host memory k_accumulate's blocks write directly: [kAccumBlocksMax][kMailboxStride] granules.  A granule
  // is ONE naturally aligned 8-byte word {evaluation tag (high), half of a double (low)} written by ONE store: the
  // value cannot arrive without its tag, whatever the fabric does with the stores of different lanes
  unsigned long long* mailbox = nullptr;
  unsigned long long mailbox_seq = 0;
  bool mailbox_check = false;     // LSA_MAILBOX_CHECK: every evaluation also folds the device-side partials and compares
  // one-launch LM solve (lsa_solve_device): granules the blocks exchange their partial sums through
  // ([2 parities][kLmBlocksMax][kMailboxStride], device memory) and the result granules in coherent host memory
  unsigned long long* lm_xchg = nullptr;
  unsigned long long* lm_mailbox = nullptr;  // [kLmMailRing][2 * kLmOut]
  unsigned lm_tag = 0;            // tags handed out so far (every launch takes max evaluations + 2)
  unsigned long long lm_seq = 0;  // launches so far
  int lm_blocks = lsa::kLmBlocks;
  // lsa_icp_gate: ICP iterations enqueued ahead of their inputs.  The host's side, in coherent host memory:
  // [kGateRing][kGateGranules] granules {sequence number of the gate, half of word i / 2} -- every 8-byte granule carries
  // its own tag, so ONE sweep of loads over the bus that finds all the tags in place has the whole block, whatever order
  // the reads were served in.  The device's side, which the launches read: [kGateRing][kGateWords] words, word 0 = go
  unsigned long long* gate_host = nullptr;
  unsigned long long* gate_dev = nullptr;
  unsigned gate_seq = 0;                 // gates enqueued so far
  int gate_current = -1;                 // ticket the launches enqueued next wait behind (lsa_match_types_gated, lsa_solve_device_begin)
  struct GateSaved { double sat[3]; int hist_pos[3]; long long hist_serial[3]; int k[3]; bool valid[3]; unsigned mask; bool used; bool link; unsigned seq; } gate_saved[lsa::kGateRing] = {};
  double* motion_dev = nullptr;          // [16] the motion within the frame (LinearTransformInterpolator) between the linked solves of one localization loop
  int debug_gate_give_up_every = 0;      // lsa_debug_set: every n-th gate gives up at once (exercises the callers' fall-back)
  int debug_lm_give_up_block = -1;       // lsa_debug_set: that workgroup of the NEXT solve abandons the exchange (one shot)
  std::deque<unsigned> lm_pending;       // result tags of the solves begun and not ended yet, oldest first
  std::deque<int> lm_pending_wait;       // ... and the gate / link each of them waits behind (-1: none)
  hipStream_t map_stream = nullptr;     // shared by the device maps of this context (lsa_device_grid.hip), created with the first of them
  int map_stream_users = 0;
  void (*solve_hook)(void*) = nullptr;  // lsa_solve_device_interlude
  void* solve_hook_arg = nullptr;
  int lm_records = 512;   // residual blocks per workgroup of the solve kernel the launch aims at (LSA_LM_RECORDS)
  int lm_cache_slots = 0;         // layers of residual blocks the solve kernel keeps in LDS (LSA_LM_CACHE caps it)
  int lm_fallbacks = 0;           // solves that timed out on the device and were redone by the host-driven loop
  // per match type a ring of kHistRing blocks of 16 ints ([8] rejection histogram + 2 hand-over counters of the kNN
  // cascade): every match takes the next block, the ring is zeroed once per turn instead of one memset per match
  int* hist_dev = nullptr;
  int hist_pos[3] = {0, 0, 0};    // block of the last match per type
  long long hist_serial[3] = {0, 0, 0};  // matches enqueued so far per type (names a block for lsa_match_histogram)
  int last_match_type = 0;
  // lanes cooperating on one query in the first kNN kernel, per keypoint type (8, 16 or 32)
  int knn_lanes[3] = {16, 8, 8};
  int knn_rounds[3] = {2, 2, 2};
  // lsa_match_types as one launch for all types, search and model fit fused (lsa_match_fused.hip); off: the staged
  // kernels of lsa_match.hip, types side by side on streams (same results, kept for comparison)
  bool fused_match = true;
  bool fused_model = true;  // ... and the search kernel fits the models of its own keypoints (one launch instead of two)
  void* trace_dev = nullptr;  // LSA_ROUTE_STATS: 4 x 8 bytes per hardware block of the last fused match (lsa_match_trace)
  bool route_stats = false;  // LSA_ROUTE_STATS: the fused search counts the routes it takes (lsa_match_route_stats)  // blocks of 3^3 .. (2 rounds + 1)^3 cells the first kernel tries
  // lsa_match_types: the keypoint types of one ICP iteration are matched concurrently, the first on
  // `stream`, the others on these, forked and joined with events (no host synchronisation)
  hipStream_t side_stream[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  void* scratch_out = nullptr;    // device staging for transformed downloads
  size_t scratch_cap = 0;
  unsigned long long* range_bits = nullptr;  // [0..1] time range, [16..24] and [32..40] bounding boxes (ordered bits)
  hipEvent_t ev_bbox = nullptr;
  hipEvent_t ev_pred = nullptr;  // the keypoints the predicted boxes are taken of exist (lsa_keypoint_boxes_predicted_mark)
  // lsa_stage_transformed: pinned host buffers the device writes the transformed keypoints into directly
  lsa_point_t* stage[3] = {nullptr, nullptr, nullptr};
  int stage_cap[3] = {0, 0, 0}, stage_n[3] = {0, 0, 0};
  hipEvent_t ev_stage = nullptr;
  bool stage_pending = false;
  bool bbox_pending = false;
  bool bbox_copied = true;   // the boxes of the last lsa_keypoint_bboxes_begin / lsa_localization_begin are on their way to the host
  bool loc_boxes = false;    // the boxes the grids read are the ones lsa_localization_begin made (words of their own)
  bool loc_armed = false;    // ... whose words are armed for the next launch
  bool pred_on_lookahead = false;  // the boxes in the first words were enqueued on the look-ahead stream (lsa_keypoint_boxes_predicted)
  std::mutex prof_mutex;  // stats / pending / event_pool of the profiling scopes
  // Buffers that were outgrown.  hipFree / hipHostFree wait for the whole device -- also for a gate that waits for THIS
  // process (lsa_icp_gate), with the runtime's lock held: a free on a worker thread at the wrong moment stalls the frame
  // until the gate gives up.  Outgrown buffers are therefore only noted here and freed at the start of the next frame
  // (lsa_collect_garbage), when nothing on the device waits for the host; kernels in flight keep the old buffer valid.
  std::mutex grave_mutex;
  std::vector<void*> grave_dev, grave_host;
  int bbox_n[3] = {0, 0, 0};

  // profiling
  bool profiling = false;
  std::string prof_only;   // when not empty: only the scope of this name is timed ...
  int prof_every = 1;      // ... and only one launch in prof_every of it (the others are counted)
  double prof_overhead_ms = 0;  // what a pair of events measures around NOTHING on this stream (calibrated when profiling is switched on): taken off every scope
  std::vector<lsa::KernelStat> stats;
  std::vector<lsa::PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;

  mutable std::mutex error_mutex;  // worker threads of the pipeline report through fail() too
  int fail(int code, const std::string& msg)
  {
    std::lock_guard<std::mutex> l(error_mutex);
    error = msg;
    return code;
  }
};

// LSA_ICP_TRACE=1: every step of the gated ICP loops on stderr (diagnostics; read once)
inline bool lsa_icp_trace_on()
{
  static const bool on = std::getenv("LSA_ICP_TRACE") != nullptr;
  return on;
}

namespace lsa
{
#define LSA_HIP(ctx, call)                                                                              \
  do                                                                                                    \
  {                                                                                                     \
    hipError_t e__ = (call);                                                                            \
    if (e__ != hipSuccess)                                                                              \
      return (ctx)->fail(LSA_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__));                \
  } while (0)

int lm_cache_capacity();
extern std::atomic<int> g_live_contexts;  // contexts of this process (lsa_lm.hip: a solve's share of the chip)
int lm_blocks_share();
inline void retire_dev(lsa_ctx* ctx, void* p)
{
  if (!p) return;
  std::lock_guard<std::mutex> l(ctx->grave_mutex);
  ctx->grave_dev.push_back(p);
}
inline void retire_host(lsa_ctx* ctx, void* p)
{
  if (!p) return;
  std::lock_guard<std::mutex> l(ctx->grave_mutex);
  ctx->grave_host.push_back(p);
}
// the 18 words of the keypoints' bounding boxes: [16..24] of range_bits for lsa_keypoint_bboxes_begin, [32..40] for lsa_localization_begin
inline unsigned* box_words(lsa_ctx* ctx) { return reinterpret_cast<unsigned*>(ctx->range_bits + 16); }
inline unsigned* loc_box_words(lsa_ctx* ctx) { return reinterpret_cast<unsigned*>(ctx->range_bits + 32); }
inline const unsigned* current_box_words(lsa_ctx* ctx) { return ctx->loc_boxes ? loc_box_words(ctx) : box_words(ctx); }
int transform_points_to(lsa_ctx* ctx, const lsa_point_t* src, int n, const double pose[16], lsa_point_t* dst, hipStream_t stream = nullptr);  // lsa_transform.hip
int transform_sets_to(lsa_ctx* ctx, const lsa_point_t* const src[3], const int n[3], const double pose[16], lsa_point_t* const dst[3], hipStream_t stream = nullptr);  // lsa_transform.hip
int time_from_advancement(lsa_ctx* ctx, lsa_point_t* frame, int n, double rpm, int first_packet);  // lsa_extract.hip
int ensure_capacity(lsa_ctx* ctx, int n);
int ensure_target(lsa_ctx* ctx, int ti, int m);
int build_target_grids(lsa_ctx* ctx, const int* tis, int count, hipStream_t st);  // lsa_match.hip: the search grids of these targets, one chain of launches
int ensure_match(lsa_ctx* ctx, int type, int k);
int ensure_scratch(lsa_ctx* ctx, size_t bytes);
int enqueue_time_range(lsa_ctx* ctx, int set, const int* counts_dev);
void finish_time_range(lsa_ctx* ctx, int set, const unsigned long long bits[2]);

// RAII helper: times the enclosed launches with HIP events on the ctx stream when profiling
struct ProfScope
{
  lsa_ctx* ctx;
  int stat = -1;
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t st = nullptr;
  ProfScope(lsa_ctx* c, const char* name, double bytes, hipStream_t stream = nullptr);
  ~ProfScope();
};
void profile_collect(lsa_ctx* ctx);
void profile_add_bytes(lsa_ctx* ctx, const char* name, double bytes);  // for scopes whose work is only known afterwards

// host-side 4x4 row-major -> Rigid
inline void row_major_to_rt(const double T[16], double R[9], double t[3])
{
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j];
    t[i] = T[i * 4 + 3];
  }
}
}  // namespace lsa
