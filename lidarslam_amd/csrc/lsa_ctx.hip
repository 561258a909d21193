// lsa_ctx.hip -- context lifetime, device buffers, frame upload / frame store,
// host-side azimuthal resolution estimate, per-kernel HIP-event profiling.
#include <algorithm>
#include <cmath>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <sched.h>
#include <string>
#include "lsa_ctx.h"
#include "../../include/lsa_pmath.h"

using namespace lsa;

namespace lsa
{

// (re)allocation of a device buffer of the context: what it held is retired, not freed (lsa_ctx.h: grave_dev)
template <typename T> static hipError_t dev_alloc(lsa_ctx* ctx, T** p, size_t count)
{
  if (*p) { retire_dev(ctx, *p); *p = nullptr; }
  return hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
}

int ensure_capacity(lsa_ctx* ctx, int n)
{
  if (n <= ctx->cap_n) return LSA_OK;
  // grow geometrically so that a sequence with slightly varying scan sizes allocates once
  int cap = std::max(n + n / 8, 4096);
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // a look-ahead extraction in flight writes into buffers that are about to move: let it finish and forget it
  if (ctx->prefetch_stream) LSA_HIP(ctx, hipStreamSynchronize(ctx->prefetch_stream));
  ctx->prefetch_pending = false;
  for (int k = 0; k < 3; ++k) LSA_HIP(ctx, dev_alloc(ctx, &ctx->kp_next[k], (size_t)cap));
  // keypoint sets must survive a growth (raw previous is still needed): copy them over
  lsa_point_t* old_kp[3][3];
  for (int s = 0; s < 3; ++s)
    for (int k = 0; k < 3; ++k) { old_kp[s][k] = ctx->kp[s][k]; ctx->kp[s][k] = nullptr; }
  for (int s = 0; s < 3; ++s)
    for (int k = 0; k < 3; ++k)
    {
      LSA_HIP(ctx, dev_alloc(ctx, &ctx->kp[s][k], (size_t)cap));
      if (old_kp[s][k] && ctx->kp_n[s][k] > 0)
        LSA_HIP(ctx, hipMemcpy(ctx->kp[s][k], old_kp[s][k], (size_t)ctx->kp_n[s][k] * sizeof(lsa_point_t), hipMemcpyDeviceToDevice));
      retire_dev(ctx, old_kp[s][k]);
    }
  {
    // an uploaded frame must survive too (a further device frame may need more room for the merged keypoints)
    lsa_point_t* old_frame = ctx->frame_own;
    const bool current = old_frame && ctx->frame == old_frame && ctx->frame_n > 0;
    ctx->frame_own = nullptr;
    LSA_HIP(ctx, dev_alloc(ctx, &ctx->frame_own, (size_t)cap));
    if (current)
    {
      LSA_HIP(ctx, hipMemcpy(ctx->frame_own, old_frame, (size_t)ctx->frame_n * sizeof(lsa_point_t), hipMemcpyDeviceToDevice));
      ctx->frame = ctx->frame_own;
    }
    retire_dev(ctx, old_frame);
  }
  LSA_HIP(ctx, dev_alloc(ctx, &ctx->xyzi, (size_t)cap));
  LSA_HIP(ctx, dev_alloc(ctx, &ctx->orig, (size_t)cap));
  LSA_HIP(ctx, dev_alloc(ctx, &ctx->ring_of, (size_t)cap));
  int nblocks = (cap + kBucketChunk - 1) / kBucketChunk;
  LSA_HIP(ctx, dev_alloc(ctx, &ctx->block_hist, (size_t)nblocks * kMaxRings));
  for (int i = 0; i < 4; ++i) LSA_HIP(ctx, dev_alloc(ctx, &ctx->score[i], (size_t)cap));
  LSA_HIP(ctx, dev_alloc(ctx, &ctx->valid, (size_t)cap));
  LSA_HIP(ctx, dev_alloc(ctx, &ctx->label, (size_t)cap));
  ctx->cap_n = cap;
  return LSA_OK;
}

int ensure_target(lsa_ctx* ctx, int ti, int m)
{
  Target& t = ctx->target[ti];
  if (!t.desc)
  {
    for (int l = 0; l < kGridLevels; ++l)
    {
      GridLevel& g = t.lv[l];
      g.max_cells = grid_level_cells(l);
      LSA_HIP(ctx, dev_alloc(ctx, &g.cell_start, (size_t)g.max_cells + 1));
      LSA_HIP(ctx, dev_alloc(ctx, &g.cell_fill, (size_t)g.max_cells));
      LSA_HIP(ctx, dev_alloc(ctx, &g.block_sums, (size_t)g.max_cells / 1024 + 2));
    }
    LSA_HIP(ctx, dev_alloc(ctx, &t.desc, kGridLevels));
    LSA_HIP(ctx, dev_alloc(ctx, &t.bbox_bits, 8));
    // armed once here, re-armed by k_grid_scatter after every build
    const int init[8] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0, 0};
    LSA_HIP(ctx, hipMemcpy(t.bbox_bits, init, sizeof(init), hipMemcpyHostToDevice));
  }
  if (m <= t.cap) return LSA_OK;
  // doubling from 64 k points (7 MB a target): a growing sub-map re-allocates a handful of times in a sequence's life -- each
  // time eight buffers are retired and freed at the next frame's start, where every hipFree waits for the device (0.3-0.6 ms:
  // with a floor of 16 k a map's first hundred keyframes crossed it for every target, 15 us a frame over bench.py's window)
  int cap = std::max(2 * m, 65536);
  // (no synchronisation: the outgrown buffers are retired, launches in flight keep them; the new ones are filled before
  // they are read.  This runs on worker threads too, beside ICP iterations that wait behind a gate)
  LSA_HIP(ctx, dev_alloc(ctx, &t.pts, (size_t)cap));
  LSA_HIP(ctx, dev_alloc(ctx, &t.xyzl, (size_t)cap));
  for (int l = 0; l < kGridLevels; ++l)
  {
    LSA_HIP(ctx, dev_alloc(ctx, &t.lv[l].sorted, (size_t)cap));
    LSA_HIP(ctx, dev_alloc(ctx, &t.lv[l].cell_of, (size_t)cap));
  }
  t.cap = cap;
  return LSA_OK;
}

int ensure_match(lsa_ctx* ctx, int type, int k)
{
  MatchBuf& b = ctx->match[type];
  if (k <= b.cap) return LSA_OK;
  int cap = std::max(k + k / 4, 4096);
  LSA_HIP(ctx, dev_alloc(ctx, &b.rec, (size_t)cap * 16));
  LSA_HIP(ctx, dev_alloc(ctx, &b.status, (size_t)cap));
  LSA_HIP(ctx, dev_alloc(ctx, &b.knn_idx, (size_t)cap * kKnnMax));
  LSA_HIP(ctx, dev_alloc(ctx, &b.knn_d2, (size_t)cap * kKnnMax));
  LSA_HIP(ctx, dev_alloc(ctx, &b.knn_cnt, (size_t)cap));
  LSA_HIP(ctx, dev_alloc(ctx, &b.slow_list, (size_t)cap));
  LSA_HIP(ctx, dev_alloc(ctx, &b.slow_pts, (size_t)cap));
  b.cap = cap;
  return LSA_OK;
}

int ensure_scratch(lsa_ctx* ctx, size_t bytes)
{
  if (bytes <= ctx->scratch_cap) return LSA_OK;
  retire_dev(ctx, ctx->scratch_out);
  ctx->scratch_out = nullptr;
  LSA_HIP(ctx, hipMalloc(&ctx->scratch_out, bytes + bytes / 4));
  ctx->scratch_cap = bytes + bytes / 4;
  return LSA_OK;
}

ProfScope::ProfScope(lsa_ctx* c, const char* name, double bytes, hipStream_t stream) : ctx(c), st(stream ? stream : c->stream)
{
  if (!ctx->profiling) return;
  // a selection names a scope or a family of scopes by their common prefix ("match_": match_search and match_model); the
  // sixty other scopes of a frame leave at once (prof_only is set while nothing is being enqueued: lsa_profile_select)
  if (!ctx->prof_only.empty() && std::strncmp(name, ctx->prof_only.c_str(), ctx->prof_only.size()) != 0) return;
  std::lock_guard<std::mutex> lock(ctx->prof_mutex);  // the device maps' insertions are enqueued (and timed) by other host threads
  for (size_t i = 0; i < ctx->stats.size(); ++i)
    if (ctx->stats[i].name == name) { stat = (int)i; break; }
  if (stat < 0)
  {
    KernelStat ks;
    ks.name = name;
    ctx->stats.push_back(ks);
    stat = (int)ctx->stats.size() - 1;
  }
  ctx->stats[stat].launches++;
  ctx->stats[stat].bytes += bytes;
  if (ctx->prof_every > 1 && (ctx->stats[stat].launches % ctx->prof_every) != 1) { stat = -1; return; }
  ctx->stats[stat].timed++;
  auto get = [&]() {
    hipEvent_t e;
    if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  };
  a = get();
  b = get();
  (void)hipEventRecord(a, st);
}
ProfScope::~ProfScope()
{
  if (stat < 0) return;
  (void)hipEventRecord(b, st);
  std::lock_guard<std::mutex> lock(ctx->prof_mutex);
  ctx->pending.push_back({stat, a, b});
}
void profile_add_bytes(lsa_ctx* ctx, const char* name, double bytes)
{
  if (!ctx->profiling) return;
  std::lock_guard<std::mutex> lock(ctx->prof_mutex);
  for (auto& st : ctx->stats)
    if (st.name == name) { st.bytes += bytes; return; }
}

void profile_collect(lsa_ctx* ctx)
{
  std::lock_guard<std::mutex> lock(ctx->prof_mutex);
  for (auto& p : ctx->pending)
  {
    (void)hipEventSynchronize(p.b);
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) ctx->stats[p.stat].total_ms += std::max(0.0, (double)ms - ctx->prof_overhead_ms);
    ctx->event_pool.push_back(p.a);
    ctx->event_pool.push_back(p.b);
  }
  ctx->pending.clear();
}

// SpinningSensorKeypointExtractor::EstimateAzimuthalResolution (SSKE.cxx:593-637), run on
// the host once, on the first usable frame: its float arithmetic (acos) feeds a threshold of
// the invalidation pass, so it must be the libm the reference itself would use.
static float estimate_azimuthal_resolution(const lsa_point_t* pts, int n)
{
  // last point seen per ring (arrival order inside a ring is the scan order)
  std::vector<int> last(kMaxRings, -1);
  std::vector<std::vector<float>> perRing(kMaxRings);
  for (int i = 0; i < n; ++i)
  {
    unsigned r = pts[i].laser_id;
    if (r >= (unsigned)kMaxRings) continue;
    if (last[r] >= 0)
    {
      const lsa_point_t& a = pts[last[r]];
      const lsa_point_t& b = pts[i];
      float d = a.x * b.x + a.y * b.y;
      float na = std::sqrt(a.x * a.x + a.y * a.y), nb = std::sqrt(b.x * b.x + b.y * b.y);
      float angle = std::abs(std::acos(d / (na * nb)));
      if (angle > 1e-4) perRing[r].push_back(angle);
    }
    last[r] = i;
  }
  std::vector<float> angles;
  angles.reserve(n);
  for (auto& v : perRing) angles.insert(angles.end(), v.begin(), v.end());
  if (angles.size() < 100) return 0.f;
  std::sort(angles.begin(), angles.end());
  unsigned maxInliersIdx = angles.size();
  float maxAngle = float(5. / 180. * M_PI);
  float medianAngle = 0.f;
  while (maxAngle > 1.8 * medianAngle)
  {
    maxInliersIdx = std::upper_bound(angles.begin(), angles.begin() + maxInliersIdx, maxAngle) - angles.begin();
    medianAngle = angles[maxInliersIdx / 2];
    maxAngle = std::min(medianAngle * 2., maxAngle / 1.8);
  }
  return medianAngle;
}

static void maybe_estimate_resolution(lsa_ctx* ctx, const lsa_point_t* pts, int n)
{
  if (ctx->az_res < 1e-6 || M_PI / 4. < ctx->az_res)
  {
    float v = estimate_azimuthal_resolution(pts, n);
    if (v > 0.f) ctx->az_res = v;
  }
}

// FNV-1a over 32 points spread evenly over the cloud (and its size): tells a buffer that was rewritten in place from the
// one that was announced.  Every sample is a cache miss in the caller's 8 MB on the frame's critical path (AddFrame
// compares before it takes the upload over): 256 samples cost 25 us a frame, 32 cost 3 -- and another scan in the same
// buffer differs in practically every point.
static unsigned long long cloud_fingerprint(const lsa_point_t* pts, int n)
{
  unsigned long long h = 1469598103934665603ull ^ (unsigned long long)n;
  const int samples = std::min(n, 32);
  for (int i = 0; i < samples; ++i)
  {
    const size_t at = (size_t)i * (size_t)n / (size_t)samples;
    unsigned long long w[sizeof(lsa_point_t) / 8];
    std::memcpy(w, pts + at, sizeof(w));
    for (unsigned long long v : w) { h ^= v; h *= 1099511628211ull; }
  }
  return h;
}

// piece `part` of the cloud being uploaded: pageable -> pinned staging -> DMA on the copy stream (any thread, any order)
static bool upload_part(lsa_ctx* ctx, const lsa_ctx::UploadSplit& u, int part)
{
  const size_t b = u.points * (size_t)part / (size_t)u.parts * sizeof(lsa_point_t), e = u.points * (size_t)(part + 1) / (size_t)u.parts * sizeof(lsa_point_t);
  if (e <= b) return true;
  std::memcpy(u.pinned + b, u.src + b, e - b);
  return hipMemcpyAsync(u.dev + b, u.pinned + b, e - b, hipMemcpyHostToDevice, ctx->copy_stream) == hipSuccess;
}
static void upload_helper_main(lsa_ctx* ctx, int part)
{
  (void)hipSetDevice(ctx->device);
  unsigned long long seen = 0;
  std::unique_lock<std::mutex> l(ctx->up_mutex);
  while (true)
  {
    ctx->up_help_cv.wait(l, [&] { return ctx->up_quit || ctx->up_split.seq != seen; });
    if (ctx->up_quit) return;
    seen = ctx->up_split.seq;
    const lsa_ctx::UploadSplit u = ctx->up_split;
    l.unlock();
    const bool ok = part < u.parts ? upload_part(ctx, u, part) : true;
    l.lock();
    if (ctx->up_split.seq == seen)  // (a helper that woke up for a cloud nobody split has nothing to report to the next one)
    {
      ctx->up_split.ok = ctx->up_split.ok && ok;
      ctx->up_split.done++;
      ctx->up_help_done.notify_all();
    }
  }
}
// the uploader thread of a context: pageable cloud -> pinned staging -> DMA on the copy stream -> event, in up_parts pieces
// side by side (this thread takes the first, a helper each of the others)
static void uploader_main(lsa_ctx* ctx)
{
  (void)hipSetDevice(ctx->device);
  std::unique_lock<std::mutex> l(ctx->up_mutex);
  while (true)
  {
    ctx->up_cv.wait(l, [ctx] { return ctx->up_quit || !ctx->up_jobs.empty(); });
    if (ctx->up_quit || ctx->up_jobs.empty()) return;  // on the way out the queued clouds are not touched: their owner may have freed them
    const int slot = ctx->up_jobs.front();
    ctx->up_jobs.pop_front();
    FrameInbox& in = ctx->inbox[slot];
    const int helpers = (int)ctx->up_helpers.size();
    lsa_ctx::UploadSplit& u = ctx->up_split;
    u.src = reinterpret_cast<const char*>(in.src); u.pinned = reinterpret_cast<char*>(in.pinned); u.dev = reinterpret_cast<char*>(in.dev);
    u.points = (size_t)in.n;
    u.parts = in.n >= 65536 ? helpers + 1 : 1;  // (a small cloud is not worth waking anybody)
    u.done = 0; u.ok = true;
    u.seq++;
    const lsa_ctx::UploadSplit mine = u;
    if (mine.parts > 1) ctx->up_help_cv.notify_all();
    l.unlock();
    bool ok = upload_part(ctx, mine, 0);
    l.lock();
    if (mine.parts > 1) ctx->up_help_done.wait(l, [&] { return ctx->up_split.done >= helpers || ctx->up_quit; });
    ok = ok && ctx->up_split.ok;
    l.unlock();
    in.fingerprint = cloud_fingerprint(in.pinned, in.n);
    ok = ok && hipEventRecord(in.ev, ctx->copy_stream) == hipSuccess;
    l.lock();
    in.state.store(ok ? 2 : -1, std::memory_order_release);
    ctx->up_done.notify_all();
  }
}

}  // namespace lsa

extern "C" {

// gives up the oldest frame uploaded ahead: its DMA has to be over before its buffers are reused
static int inbox_drop_front(lsa_ctx* ctx)
{
  FrameInbox& old = ctx->inbox[ctx->inbox_queue.front()];
  {
    std::unique_lock<std::mutex> l(ctx->up_mutex);
    ctx->up_done.wait(l, [&] { return old.state.load() != 1; });
  }
  if (old.state.load() == 2) LSA_HIP(ctx, hipEventSynchronize(old.ev));
  if (ctx->prefetch_pending && ctx->prefetch_frame == old.dev)
  {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->prefetch_stream));
    ctx->prefetch_pending = false;
  }
  old.state.store(0);
  ctx->inbox_queue.pop_front();
  return LSA_OK;
}

int lsa_upload_frame_begin(lsa_ctx* ctx, const lsa_point_t* pts, int n)
{
  if (!ctx || !pts || n <= 0) return ctx ? ctx->fail(LSA_E_ARG, "lsa_upload_frame_begin: empty frame") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  // two frames ahead at most: the cloud of the next AddFrame (announced during the previous one) and the one after it
  while (ctx->inbox_queue.size() >= 2)
  {
    const int rc = inbox_drop_front(ctx);
    if (rc) return rc;
  }
  int slot = -1;
  for (int c = 0; c < 3 && slot < 0; ++c)
  {
    bool used = c == ctx->inbox_current;
    for (int q : ctx->inbox_queue) used = used || q == c;
    if (!used) slot = c;
  }
  if (slot < 0) return ctx->fail(LSA_E_STATE, "lsa_upload_frame_begin: no free buffer");
  FrameInbox& in = ctx->inbox[slot];
  if (!ctx->uploader.joinable())
  {
    if (const char* e = std::getenv("LSA_UPLOAD_THREADS")) ctx->up_parts = std::min(std::max(std::atoi(e), 1), 8);
    for (int h = 1; h < ctx->up_parts; ++h) ctx->up_helpers.emplace_back(upload_helper_main, ctx, h);
    ctx->uploader = std::thread(uploader_main, ctx);
  }
  if (!in.ev) LSA_HIP(ctx, hipEventCreateWithFlags(&in.ev, hipEventDisableTiming));
  if (in.cap < n)
  {
    // (this slot's last frame is at least two AddFrame calls old: nothing reads it any more)
    retire_dev(ctx, in.dev);
    retire_host(ctx, in.pinned);
    in.dev = nullptr; in.pinned = nullptr; in.cap = 0;
    const int cap = n + n / 8;
    LSA_HIP(ctx, hipMalloc((void**)&in.dev, (size_t)cap * sizeof(lsa_point_t)));
    LSA_HIP(ctx, hipHostMalloc((void**)&in.pinned, (size_t)cap * sizeof(lsa_point_t), hipHostMallocDefault));
    in.cap = cap;
  }
  in.n = n;
  in.src = pts;
  in.state.store(1, std::memory_order_release);
  ctx->inbox_queue.push_back(slot);
  {
    std::lock_guard<std::mutex> l(ctx->up_mutex);
    ctx->up_jobs.push_back(slot);
  }
  ctx->up_cv.notify_one();
  return LSA_OK;
}

int lsa_upload_frame_ready(const lsa_ctx* ctx)
{
  if (!ctx || ctx->inbox_queue.empty()) return 0;
  return ctx->inbox[ctx->inbox_queue.front()].state.load(std::memory_order_acquire) == 2 ? 1 : 0;
}

int lsa_upload_frame_adopt(lsa_ctx* ctx, const lsa_point_t* pts, int n)
{
  if (!ctx) return LSA_E_ARG;
  // the announced cloud this one is, if any (clouds announced before it were skipped by the caller: given up)
  size_t at = ctx->inbox_queue.size();
  for (size_t i = 0; i < ctx->inbox_queue.size() && at == ctx->inbox_queue.size(); ++i)
    if (ctx->inbox[ctx->inbox_queue[i]].src == pts && ctx->inbox[ctx->inbox_queue[i]].n == n) at = i;
  if (at == ctx->inbox_queue.size()) return 0;  // not announced: the caller uploads this one itself
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  for (size_t i = 0; i < at; ++i)
  {
    const int rc = inbox_drop_front(ctx);
    if (rc) return rc;
  }
  const int slot = ctx->inbox_queue.front();
  FrameInbox& in = ctx->inbox[slot];
  {
    std::unique_lock<std::mutex> l(ctx->up_mutex);
    ctx->up_done.wait(l, [&] { return in.state.load() != 1; });
  }
  ctx->inbox_queue.pop_front();
  if (in.state.load() != 2)
  {
    in.state.store(0);
    return ctx->fail(LSA_E_HIP, "lsa_upload_frame_adopt: the upload failed");
  }
  if (in.fingerprint != cloud_fingerprint(pts, n))
  {
    // same address and size, other contents: the buffer was reused for another scan since it was announced (a driver's
    // ring buffer, an allocator handing the same block out again) -- the copy made then is stale, the caller uploads
    LSA_HIP(ctx, hipEventSynchronize(in.ev));
    if (ctx->prefetch_pending && ctx->prefetch_frame == in.dev)
    {
      LSA_HIP(ctx, hipStreamSynchronize(ctx->prefetch_stream));
      ctx->prefetch_pending = false;
    }
    in.state.store(0);
    return 0;
  }
  int rc = ensure_capacity(ctx, n);
  if (rc) return rc;
  maybe_estimate_resolution(ctx, pts, n);
  LSA_HIP(ctx, hipStreamWaitEvent(ctx->stream, in.ev, 0));
  ctx->frame = in.dev;
  ctx->frame_n = n;
  ctx->inbox_current = slot;
  in.state.store(0);
  ctx->uploads_adopted++;
  return 1;
}

int lsa_upload_frame_forget(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  while (!ctx->inbox_queue.empty())
  {
    const int rc = inbox_drop_front(ctx);
    if (rc) return rc;
  }
  return LSA_OK;
}

int lsa_pin_host_memory(void* ptr, size_t bytes)
{
  if (!ptr || bytes == 0) return LSA_E_ARG;
  return hipHostRegister(ptr, bytes, hipHostRegisterPortable) == hipSuccess ? LSA_OK : LSA_E_HIP;
}
int lsa_unpin_host_memory(void* ptr)
{
  if (!ptr) return LSA_E_ARG;
  return hipHostUnregister(ptr) == hipSuccess ? LSA_OK : LSA_E_HIP;
}

int lsa_collect_garbage(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  std::vector<void*> dev, host;
  {
    std::lock_guard<std::mutex> l(ctx->grave_mutex);
    dev.swap(ctx->grave_dev);
    host.swap(ctx->grave_host);
  }
  if (dev.empty() && host.empty()) return LSA_OK;
  if (std::getenv("LSA_STAGE_DEBUG")) std::fprintf(stderr, "[garbage debug] %zu device and %zu host buffers freed\n", dev.size(), host.size());
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  for (void* p : dev) (void)hipFree(p);       // (waits for the device: every launch that could still read them is over)
  for (void* p : host) (void)hipHostFree(p);
  return LSA_OK;
}

int lsa_uploads_adopted(const lsa_ctx* ctx) { return ctx ? ctx->uploads_adopted : 0; }

int lsa_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int lsa_ctx_create(int device_id, lsa_ctx** out)
{
  if (!out) return LSA_E_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n)
    return LSA_E_NO_DEVICE;  // no CPU fallback: the caller must fail loudly
  if (hipSetDevice(device_id) != hipSuccess) return LSA_E_HIP;
  lsa_ctx* ctx = new lsa_ctx;
  ctx->device = device_id;
  g_live_contexts.fetch_add(1, std::memory_order_relaxed);
  // EXACTLY THREE streams per context, created together: the registration's, the look-ahead's, the copies'.  The runtime
  // deals streams to the process's four hardware queues in creation order (GPU_MAX_HW_QUEUES = 4, round robin), and two
  // streams on one queue wait for each other's kernels.  Three in a row sit on three different queues, and the next
  // context's three start one queue further: with four streams per context every context's registration stream landed
  // on the same queue (8 sequences side by side: 1 200 frames/s against 2 000), and any further stream of a context
  // shares the registration's queue (a stream for the maps: 765 against 890 frames/s for one sequence).  The side streams
  // of the staged (non-fused) match are created when that path is first used.
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { g_live_contexts.fetch_sub(1, std::memory_order_relaxed); delete ctx; return LSA_E_HIP; }
  bool ok = true;
  ok &= hipStreamCreateWithFlags(&ctx->prefetch_stream, hipStreamNonBlocking) == hipSuccess;
  ok &= hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 2; ++i) ok &= hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) == hipSuccess;
  ok &= hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
  ok &= hipEventCreateWithFlags(&ctx->ev_bbox, hipEventDisableTiming) == hipSuccess;
  ok &= hipEventCreateWithFlags(&ctx->ev_pred, hipEventDisableTiming) == hipSuccess;
  ok &= hipEventCreateWithFlags(&ctx->ev_stage, hipEventDisableTiming) == hipSuccess;
  ok &= hipMalloc((void**)&ctx->ring_start, (kMaxRings + 1) * sizeof(int)) == hipSuccess;
  ok &= hipMalloc((void**)&ctx->ring_len, kMaxRings * sizeof(int)) == hipSuccess;
  ok &= hipMalloc((void**)&ctx->extract_out, 16 * sizeof(int)) == hipSuccess;
  ok &= hipMalloc((void**)&ctx->extract_out_next, 16 * sizeof(int)) == hipSuccess;
  ok &= hipHostMalloc((void**)&ctx->host_next, 16 * sizeof(int), hipHostMallocDefault) == hipSuccess;
  ok &= hipEventCreateWithFlags(&ctx->ev_prefetch, hipEventDisableTiming) == hipSuccess;
  ok &= hipEventCreateWithFlags(&ctx->ev_spare, hipEventDisableTiming) == hipSuccess;
  for (int k = 0; k < 3; ++k) ok &= hipEventCreateWithFlags(&ctx->ev_map_ahead[k], hipEventDisableTiming) == hipSuccess;
  ok &= hipEventCreateWithFlags(&ctx->ev_kp_ready, hipEventDisableTiming) == hipSuccess;
  if (ok) { ctx->kp_count_dev = ctx->extract_out; ctx->ring_meta = ctx->extract_out + 4; }
  ok &= hipMalloc((void**)&ctx->ring_counts, kMaxRings * 3 * sizeof(int)) == hipSuccess;
  ok &= hipMalloc((void**)&ctx->partials, (size_t)kAccumBlocksMax * kAccumVals * sizeof(double)) == hipSuccess;
  ok &= hipMalloc((void**)&ctx->reduce_out, 64 * sizeof(double)) == hipSuccess;
  if (ok) ok &= hipMemset(ctx->reduce_out, 0, 64 * sizeof(double)) == hipSuccess;  // [32] holds the arrival ticket of k_accumulate
  ok &= hipMalloc((void**)&ctx->hist_dev, 3 * kHistRing * 16 * sizeof(int)) == hipSuccess;
  if (ok) ok &= hipMemset(ctx->hist_dev, 0, 3 * kHistRing * 16 * sizeof(int)) == hipSuccess;
  ok &= hipMalloc((void**)&ctx->range_bits, 48 * sizeof(unsigned long long)) == hipSuccess;
  ok &= hipHostMalloc((void**)&ctx->host_pinned, 512 * sizeof(double), hipHostMallocDefault) == hipSuccess;
  if (hipHostMalloc((void**)&ctx->mailbox, (size_t)kAccumBlocksMax * kMailboxStride * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess)
    std::memset(ctx->mailbox, 0, (size_t)kAccumBlocksMax * kMailboxStride * sizeof(unsigned long long));
  else
    ctx->mailbox = nullptr;  // optional: lsa_accumulate falls back to a copy + synchronise
  if (hipHostMalloc((void**)&ctx->lm_mailbox, (size_t)kLmMailRing * 2 * kLmOut * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess)
    std::memset(ctx->lm_mailbox, 0, (size_t)kLmMailRing * 2 * kLmOut * sizeof(unsigned long long));
  else
    ctx->lm_mailbox = nullptr;  // optional: lsa_solve_device then reports LSA_E_STATE and the host-driven loop is used
  // gates of ICP iterations enqueued ahead (lsa_icp_gate): optional like the result mailbox
  if (ctx->lm_mailbox && hipHostMalloc((void**)&ctx->gate_host, (size_t)kGateRing * kGateGranules * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess)
  {
    std::memset(ctx->gate_host, 0, (size_t)kGateRing * kGateGranules * sizeof(unsigned long long));
    if (hipMalloc((void**)&ctx->gate_dev, (size_t)kGateRing * kGateWords * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(ctx->gate_dev, 0, (size_t)kGateRing * kGateWords * sizeof(unsigned long long)) != hipSuccess)
    {
      (void)hipHostFree(ctx->gate_host);
      ctx->gate_host = nullptr;
      ctx->gate_dev = nullptr;
    }
  }
  else
    ctx->gate_host = nullptr;
  if (ctx->gate_dev && (hipMalloc((void**)&ctx->motion_dev, 16 * sizeof(double)) != hipSuccess || hipMemset(ctx->motion_dev, 0, 16 * sizeof(double)) != hipSuccess))
    ctx->motion_dev = nullptr;  // optional: no links then (lsa_icp_link says so)
  ok &= hipMalloc((void**)&ctx->lm_xchg, (size_t)2 * kLmBlocksMax * kMailboxStride * sizeof(unsigned long long)) == hipSuccess;
  if (ok) ok &= hipMemset(ctx->lm_xchg, 0, (size_t)2 * kLmBlocksMax * kMailboxStride * sizeof(unsigned long long)) == hipSuccess;
  if (const char* e = std::getenv("LSA_ACCUM_BLOCKS")) ctx->accum_blocks = std::min(std::max(std::atoi(e), 1), kAccumBlocksMax);
  ctx->lm_cache_slots = lm_cache_capacity();
  if (const char* e = std::getenv("LSA_LM_CACHE")) ctx->lm_cache_slots = std::min(std::max(std::atoi(e), 0), ctx->lm_cache_slots);
  if (const char* e = std::getenv("LSA_LM_BLOCKS")) ctx->lm_blocks = std::min(std::max(std::atoi(e), 1), kLmBlocksMax);
  if (const char* e = std::getenv("LSA_LM_RECORDS")) ctx->lm_records = std::min(std::max(std::atoi(e), 256), 4096);
  if (const char* e = std::getenv("LSA_ROUTE_STATS")) ctx->route_stats = std::atoi(e) != 0;
  if (ctx->route_stats) ok &= hipMalloc(&ctx->trace_dev, ((size_t)8192 * 12 + 16) * sizeof(unsigned long long)) == hipSuccess && hipMemset(ctx->trace_dev, 0, ((size_t)8192 * 12 + 16) * sizeof(unsigned long long)) == hipSuccess;
  if (const char* e = std::getenv("LSA_FUSED_MATCH")) ctx->fused_match = std::atoi(e) != 0;
  if (const char* e = std::getenv("LSA_FUSED_MODEL")) ctx->fused_model = std::atoi(e) != 0;
  if (const char* e = std::getenv("LSA_MAILBOX_CHECK")) ctx->mailbox_check = std::atoi(e) != 0;
  if (!ok) { lsa_ctx_destroy(ctx); return LSA_E_HIP; }
  *out = ctx;
  return LSA_OK;
}

void lsa_ctx_destroy(lsa_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < 2; ++i)
    if (ctx->side_stream[i]) (void)hipStreamSynchronize(ctx->side_stream[i]);
  if (ctx->prefetch_stream) (void)hipStreamSynchronize(ctx->prefetch_stream);
  if (ctx->uploader.joinable())
  {
    {
      std::lock_guard<std::mutex> l(ctx->up_mutex);
      ctx->up_quit = true;
    }
    ctx->up_cv.notify_all();
    ctx->up_help_cv.notify_all();
    ctx->up_help_done.notify_all();
    ctx->uploader.join();
    for (auto& h : ctx->up_helpers) h.join();
    ctx->up_helpers.clear();
  }
  (void)lsa_collect_garbage(ctx);
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
  for (auto& in : ctx->inbox)
  {
    if (in.dev) (void)hipFree(in.dev);
    if (in.pinned) (void)hipHostFree(in.pinned);
    if (in.ev) (void)hipEventDestroy(in.ev);
  }
  profile_collect(ctx);
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  fr(ctx->frame_own); fr(ctx->xyzi); fr(ctx->orig); fr(ctx->ring_of); fr(ctx->block_hist);
  fr(ctx->ring_start); fr(ctx->ring_len); fr(ctx->extract_out); fr(ctx->extract_out_next);
  for (int k = 0; k < 3; ++k) fr(ctx->kp_next[k]);
  if (ctx->host_next) (void)hipHostFree(ctx->host_next);
  if (ctx->ev_prefetch) (void)hipEventDestroy(ctx->ev_prefetch);
  if (ctx->ev_spare) (void)hipEventDestroy(ctx->ev_spare);
  for (int k = 0; k < 3; ++k)
    if (ctx->ev_map_ahead[k]) (void)hipEventDestroy(ctx->ev_map_ahead[k]);
  if (ctx->ev_kp_ready) (void)hipEventDestroy(ctx->ev_kp_ready);
  if (ctx->prefetch_stream) (void)hipStreamDestroy(ctx->prefetch_stream);
  for (int i = 0; i < 4; ++i) fr(ctx->score[i]);
  fr(ctx->valid); fr(ctx->label); fr(ctx->ring_counts);
  for (int s = 0; s < 3; ++s) for (int k = 0; k < 3; ++k) fr(ctx->kp[s][k]);
  for (int k = 0; k < 12; ++k)
  {
    Target& t = ctx->target[k];
    fr(t.pts); fr(t.xyzl); fr(t.desc); fr(t.bbox_bits);
    for (int l = 0; l < kGridLevels; ++l) { fr(t.lv[l].sorted); fr(t.lv[l].cell_of); fr(t.lv[l].cell_start); fr(t.lv[l].cell_fill); fr(t.lv[l].block_sums); }
  }
  for (int k = 0; k < 3; ++k)
  {
    fr(ctx->match[k].rec); fr(ctx->match[k].status); fr(ctx->match[k].knn_idx); fr(ctx->match[k].knn_d2); fr(ctx->match[k].knn_cnt); fr(ctx->match[k].slow_list); fr(ctx->match[k].slow_pts);
  }
  fr(ctx->partials); fr(ctx->reduce_out); fr(ctx->hist_dev); fr(ctx->scratch_out); fr(ctx->range_bits);
  for (auto& s : ctx->store) fr(s.first);
  if (ctx->host_pinned) (void)hipHostFree(ctx->host_pinned);
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  if (ctx->lm_mailbox) (void)hipHostFree(ctx->lm_mailbox);
  if (ctx->gate_host) (void)hipHostFree(ctx->gate_host);
  fr(ctx->gate_dev);
  fr(ctx->motion_dev);
  fr(ctx->lm_xchg);
  fr(ctx->trace_dev);
  for (int i = 0; i < 2; ++i)
  {
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
    if (ctx->side_stream[i]) (void)hipStreamDestroy(ctx->side_stream[i]);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_bbox) (void)hipEventDestroy(ctx->ev_bbox);
  if (ctx->ev_pred) (void)hipEventDestroy(ctx->ev_pred);
  if (ctx->ev_stage) (void)hipEventDestroy(ctx->ev_stage);
  for (int k = 0; k < 3; ++k)
    if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
  for (int k = 0; k < 6; ++k)
    if (ctx->tstage[k]) (void)hipHostFree(ctx->tstage[k]);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  g_live_contexts.fetch_sub(1, std::memory_order_relaxed);
  delete ctx;
}

const char* lsa_last_error(const lsa_ctx* ctx)
{
  if (!ctx) return "null context";
  // a copy of the calling thread's own: another thread of the pipeline may report an error meanwhile
  thread_local std::string copy;
  {
    std::lock_guard<std::mutex> l(ctx->error_mutex);
    copy = ctx->error;
  }
  return copy.c_str();
}

int lsa_sync(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSA_OK;
}

// ---- the driver's wire format straight to the device --------------------------------------
namespace
{
struct WireMap
{
  int advancement;  // 1: the time field receives the azimuth advancement in [0, 1) instead of the record's time
  lsa_wire_layout_t lay;
  int mapping_len;
  int device_id;
  uint16_t mapping[kMaxRings];
};
// one LidarPoint per wire record (VelodyneToLidarNode.cxx:81-96): coordinates, intensity, mapped ring, device,
// time offset widened to double
__global__ __launch_bounds__(256) void k_wire_to_points(const unsigned char* __restrict__ raw, int n, WireMap m, float4* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char* r = raw + (size_t)i * m.lay.point_step;
  auto f32 = [&](int off) { float v; memcpy(&v, r + off, sizeof(v)); return v; };
  uint16_t ring;
  memcpy(&ring, r + m.lay.off_ring, sizeof(ring));
  const unsigned id = m.mapping_len > 0 ? (ring < m.mapping_len ? m.mapping[ring] : 0xffffu) : ring;
  double t = (double)f32(m.lay.off_time);
  if (m.advancement)
  {
    // SpinningFrameAdvancementEstimator (lidar_conversions/src/Utilities.h:88-100): the azimuth of the point as a
    // fraction of a turn, relative to the frame's first point, wrapped into [0, 1).  std::fmod is exact, and so is
    // x - trunc(x) for |x| < 2^52: wrap() below IS std::fmod(1 + std::fmod(x, 1), 1).  The arc tangent is the portable
    // one evaluated in double and rounded to float (the node calls the float overload).
    auto adv_of = [&](const unsigned char* rr) {
      float x, y;
      memcpy(&x, rr + m.lay.off_x, sizeof(x));
      memcpy(&y, rr + m.lay.off_y, sizeof(y));
      return (3.14159265358979323846 - (double)(float)lsa_atan2((double)y, (double)x)) / (2 * 3.14159265358979323846);
    };
    auto wrap = [](double x) { const double f = x - trunc(x); const double g = 1.0 + f; return g - trunc(g); };
    t = wrap(adv_of(r) - adv_of(raw));
  }
  const long long tb = __double_as_longlong(t);
  float4 a = make_float4(f32(m.lay.off_x), f32(m.lay.off_y), f32(m.lay.off_z), 1.f);
  float4 b;
  b.x = __int_as_float((int)(tb & 0xffffffffll));
  b.y = __int_as_float((int)(tb >> 32));
  b.z = f32(m.lay.off_intensity);
  b.w = __uint_as_float(id | ((unsigned)(m.device_id & 0xff) << 16));  // laser_id u16, device_id u8, label u8 = 0
  out[2 * (size_t)i] = a;
  out[2 * (size_t)i + 1] = b;
}
// lidar_conversions::Utils::SpinningFrameAdvancementEstimator (ros_wrapping/lidar_conversions/src/Utilities.h:62-114)
struct FrameAdvancementEstimator
{
  double init = 0.;
  bool first = true;
  std::vector<double> prev = std::vector<double>(65536, 0.);  // std::map<int, double>: a missing ring reads 0
  double operator()(float x, float y, unsigned laser_id)
  {
    const double adv = (M_PI - std::atan2(y, x)) / (2 * M_PI);
    if (first) { init = adv; first = false; }
    auto wrapMax = [](double v, double max) { return std::fmod(max + std::fmod(v, max), max); };
    double frameAdv = wrapMax(adv - init, 1.);
    if (frameAdv < prev[laser_id]) frameAdv += 1.;
    prev[laser_id] = frameAdv;
    return frameAdv;
  }
};
}  // namespace

int lsa_upload_wire_frame(lsa_ctx* ctx, const void* data, int n, const lsa_wire_layout_t* lay, const uint16_t* laser_id_mapping, int mapping_len,
                          int device_id, double rpm, int timestamp_first_packet)
{
  if (!ctx || !data || !lay || n <= 0 || lay->point_step <= 0 || mapping_len < 0 || (mapping_len > 0 && !laser_id_mapping))
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_upload_wire_frame: empty frame or bad layout") : LSA_E_ARG;
  if (mapping_len > kMaxRings) return ctx->fail(LSA_E_CAPACITY, "lsa_upload_wire_frame: more than 512 entries in the laser id mapping");
  const int offs[6] = {lay->off_x, lay->off_y, lay->off_z, lay->off_intensity, lay->off_time, lay->off_ring};
  for (int i = 0; i < 6; ++i)
    if (offs[i] < 0 || offs[i] + (i == 5 ? 2 : 4) > lay->point_step) return ctx->fail(LSA_E_ARG, "lsa_upload_wire_frame: field outside the record");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  const unsigned char* raw = static_cast<const unsigned char*>(data);
  auto f32 = [&](int i, int off) { float v; std::memcpy(&v, raw + (size_t)i * lay->point_step + off, sizeof(v)); return v; };
  // "If first and last points have same timestamps, this is not normal" (VelodyneToLidarNode.cxx:74)
  const bool isTimeValid = f32(n - 1, lay->off_time) - f32(0, lay->off_time) > 1e-8;
  auto on_host = [&]() -> int {
    // host conversion (libm atan2 / fmod, ring by ring in arrival order, as the driver node does): the first frame,
    // whose azimuthal resolution is estimated on the host from the converted points anyway, and frames the device
    // cannot bucket by ring
    std::vector<lsa_point_t> pts(n);
    FrameAdvancementEstimator est;
    for (int i = 0; i < n; ++i)
    {
      lsa_point_t& p = pts[i];
      uint16_t ring;
      std::memcpy(&ring, raw + (size_t)i * lay->point_step + lay->off_ring, sizeof(ring));
      p.x = f32(i, lay->off_x); p.y = f32(i, lay->off_y); p.z = f32(i, lay->off_z); p.w = 1.f;
      p.intensity = f32(i, lay->off_intensity);
      p.laser_id = mapping_len > 0 ? (ring < mapping_len ? laser_id_mapping[ring] : (uint16_t)0xffff) : ring;
      p.device_id = (uint8_t)device_id;
      p.label = 0;
      if (isTimeValid) p.time = f32(i, lay->off_time);
      else
      {
        const double adv = est(p.x, p.y, p.laser_id);
        p.time = (timestamp_first_packet ? adv : adv - 1) / rpm * 60.;
      }
    }
    const int rc = lsa_upload_frame(ctx, pts.data(), n);
    if (rc) return rc;
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pts goes away
    return LSA_OK;
  };
  if (ctx->az_res <= 0.f) return on_host();
  int rc = ensure_capacity(ctx, n);
  if (rc) return rc;
  const size_t bytes = (size_t)n * lay->point_step;
  rc = ensure_scratch(ctx, bytes);
  if (rc) return rc;
  WireMap m;
  m.advancement = isTimeValid ? 0 : 1;
  m.lay = *lay;
  m.mapping_len = mapping_len;
  m.device_id = device_id;
  if (mapping_len > 0) std::memcpy(m.mapping, laser_id_mapping, (size_t)mapping_len * sizeof(uint16_t));
  LSA_HIP(ctx, hipMemcpyAsync(ctx->scratch_out, data, bytes, hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps(ctx, "wire_to_points", (double)n * (lay->point_step + 32));
    hipLaunchKernelGGL(k_wire_to_points, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, static_cast<const unsigned char*>(ctx->scratch_out), n, m,
                       reinterpret_cast<float4*>(ctx->frame_own));
  }
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's buffer may be pageable and go away
  if (!isTimeValid)
  {
    // no usable time field: built from the azimuth advancement (a per-ring "first descent" found on the ring-bucketed
    // frame), on the device; laser ids the bucketing cannot hold go through the host
    rc = time_from_advancement(ctx, ctx->frame_own, n, rpm, timestamp_first_packet);
    if (rc == LSA_E_CAPACITY) return on_host();
    if (rc) return rc;
  }
  ctx->frame = ctx->frame_own;
  ctx->frame_n = n;
  ctx->inbox_current = -1;
  return LSA_OK;
}

int lsa_upload_frame(lsa_ctx* ctx, const lsa_point_t* pts, int n)
{
  if (!ctx || !pts || n <= 0) return ctx ? ctx->fail(LSA_E_ARG, "lsa_upload_frame: empty frame") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_capacity(ctx, n);
  if (rc) return rc;
  maybe_estimate_resolution(ctx, pts, n);
  LSA_HIP(ctx, hipMemcpyAsync(ctx->frame_own, pts, (size_t)n * sizeof(lsa_point_t), hipMemcpyHostToDevice, ctx->stream));
  ctx->frame = ctx->frame_own;
  ctx->frame_n = n;
  ctx->inbox_current = -1;
  return LSA_OK;
}

// ---- vtkSlam::PolyDataToPointCloud (paraview_wrapping/Plugin/vtkLidarSlam/vtkSlam.cxx:668-707) on the device -----------
namespace
{
struct SoaFrame
{
  const void* xyz; const void* time; const void* laser; const void* intensity;
  int xyz_type, time_type, laser_type, intensity_type;
  int n;
  int mapping_len;
  double factor;  // TimeToSecondsFactor
  uint16_t mapping[kMaxRings];
};
__device__ __forceinline__ double soa_value(const void* base, int type, size_t i)
{
  switch (type)
  {
    case LSA_SCALAR_F32: return (double)static_cast<const float*>(base)[i];
    case LSA_SCALAR_F64: return static_cast<const double*>(base)[i];
    case LSA_SCALAR_U8: return (double)static_cast<const unsigned char*>(base)[i];
    case LSA_SCALAR_U16: return (double)static_cast<const unsigned short*>(base)[i];
    case LSA_SCALAR_U32: return (double)static_cast<const unsigned int*>(base)[i];
    default: return (double)static_cast<const int*>(base)[i];
  }
}
__device__ __forceinline__ long long ordered_bits(double v)
{
  const long long b = __double_as_longlong(v);
  return b >= 0 ? b : b ^ 0x7fffffffffffffffll;
}
// frame end time = the largest value of the time array (arrayTime->GetRange()[1]); points with all-zero coordinates
// are dropped: per chunk of 1024 points, how many stay
__global__ __launch_bounds__(256) void k_soa_scan_chunks(SoaFrame f, long long* __restrict__ tmax, int* __restrict__ chunk_count)
{
  __shared__ int cnt[4];
  __shared__ long long mx[4];
  int mine = 0;
  long long m = (long long)0x8000000000000000ull;
  for (int q = 0; q < 4; ++q)
  {
    const int i = blockIdx.x * 1024 + q * 256 + threadIdx.x;
    if (i < f.n)
    {
      const double x = soa_value(f.xyz, f.xyz_type, 3 * (size_t)i), y = soa_value(f.xyz, f.xyz_type, 3 * (size_t)i + 1), z = soa_value(f.xyz, f.xyz_type, 3 * (size_t)i + 2);
      if (x != 0. || y != 0. || z != 0.) ++mine;
      const long long t = ordered_bits(soa_value(f.time, f.time_type, i));
      m = t > m ? t : m;
    }
  }
  for (int o = 32; o > 0; o >>= 1)
  {
    mine += __shfl_down(mine, o);
    const long long t = __shfl_down(m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) { cnt[threadIdx.x >> 6] = mine; mx[threadIdx.x >> 6] = m; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    chunk_count[blockIdx.x] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    long long a = mx[0] > mx[1] ? mx[0] : mx[1], b = mx[2] > mx[3] ? mx[2] : mx[3];
    atomicMax(tmax, a > b ? a : b);
  }
}
// exclusive scan of the chunk counts (one block; a frame has a few hundred chunks)
__global__ __launch_bounds__(1024) void k_soa_scan_counts(int* __restrict__ chunk_count, int nchunks, int* __restrict__ total)
{
  __shared__ int s[1024];
  int run = 0;
  for (int base = 0; base < nchunks; base += 1024)
  {
    const int i = base + threadIdx.x;
    const int v = i < nchunks ? chunk_count[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1)
    {
      const int a = threadIdx.x >= (unsigned)o ? s[threadIdx.x - o] : 0;
      __syncthreads();
      s[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < nchunks) chunk_count[i] = run + s[threadIdx.x] - v;
    run += s[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = run;
}
// the points that stay, in order, as LidarPoints: time relative to the frame's end [s], laser id (mapped), intensity
__global__ __launch_bounds__(256) void k_soa_to_points(SoaFrame f, const long long* __restrict__ tmax, const int* __restrict__ chunk_start, float4* __restrict__ out)
{
  __shared__ int wave_base[4];
  const long long tb = *tmax;
  const double end_time = __longlong_as_double(tb >= 0 ? tb : tb ^ 0x7fffffffffffffffll);
  int run = chunk_start[blockIdx.x];
  for (int q = 0; q < 4; ++q)
  {
    const int i = blockIdx.x * 1024 + q * 256 + threadIdx.x;
    double x = 0., y = 0., z = 0.;
    if (i < f.n)
    {
      x = soa_value(f.xyz, f.xyz_type, 3 * (size_t)i); y = soa_value(f.xyz, f.xyz_type, 3 * (size_t)i + 1); z = soa_value(f.xyz, f.xyz_type, 3 * (size_t)i + 2);
    }
    const bool keep = i < f.n && (x != 0. || y != 0. || z != 0.);
    const unsigned long long ballot = __ballot(keep);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wave_base[wv] = __popcll(ballot);
    __syncthreads();
    int base = run;
    for (int w = 0; w < wv; ++w) base += wave_base[w];
    const int batch = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
    if (keep)
    {
      const int at = base + __popcll(ballot & ((1ull << lane) - 1ull));
      const double t = (soa_value(f.time, f.time_type, i) - end_time) * f.factor;
      const double lid = soa_value(f.laser, f.laser_type, i);
      unsigned id = f.mapping_len > 0 ? ((size_t)lid < (size_t)f.mapping_len ? f.mapping[(size_t)lid] : 0xffffu) : (unsigned)(unsigned short)lid;
      const long long bits = __double_as_longlong(t);
      float4 a = make_float4((float)x, (float)y, (float)z, 1.f), b;
      b.x = __int_as_float((int)(bits & 0xffffffffll));
      b.y = __int_as_float((int)(bits >> 32));
      b.z = (float)soa_value(f.intensity, f.intensity_type, i);
      b.w = __uint_as_float(id & 0xffffu);  // device_id 0, label 0
      out[2 * (size_t)at] = a;
      out[2 * (size_t)at + 1] = b;
    }
    run += batch;
    __syncthreads();
  }
}
int scalar_size(int type) { return type == LSA_SCALAR_F64 ? 8 : type == LSA_SCALAR_U8 ? 1 : type == LSA_SCALAR_U16 ? 2 : 4; }
}  // namespace

int lsa_upload_polydata_frame(lsa_ctx* ctx, int n, const void* xyz, int xyz_type, const void* time, int time_type, const void* laser_id, int laser_type,
                              const void* intensity, int intensity_type, const uint16_t* laser_id_mapping, int mapping_len, double time_to_seconds,
                              uint64_t* stamp_us, int* n_valid)
{
  if (!ctx || n <= 0 || !xyz || !time || !laser_id || !intensity || mapping_len < 0 || (mapping_len > 0 && !laser_id_mapping) ||
      (xyz_type != LSA_SCALAR_F32 && xyz_type != LSA_SCALAR_F64))
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_upload_polydata_frame: bad argument") : LSA_E_ARG;
  for (int t : {time_type, laser_type, intensity_type})
    if (t < LSA_SCALAR_F32 || t > LSA_SCALAR_I32) return ctx->fail(LSA_E_ARG, "lsa_upload_polydata_frame: unknown scalar type");
  if (mapping_len > kMaxRings) return ctx->fail(LSA_E_CAPACITY, "lsa_upload_polydata_frame: more than 512 entries in the laser id mapping");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_capacity(ctx, n);
  if (rc) return rc;
  // the four arrays go to the device as they are (structure of arrays, no LidarPoint cloud is built on the host)
  const size_t sz[4] = {(size_t)3 * n * scalar_size(xyz_type), (size_t)n * scalar_size(time_type), (size_t)n * scalar_size(laser_type),
                        (size_t)n * scalar_size(intensity_type)};
  size_t off[4], total = 0;
  for (int i = 0; i < 4; ++i) { off[i] = total; total += (sz[i] + 255) / 256 * 256; }
  const int nchunks = (n + 1023) / 1024;
  const size_t off_counts = total, off_tmax = off_counts + ((size_t)(nchunks + 1) * sizeof(int) + 255) / 256 * 256;
  rc = ensure_scratch(ctx, off_tmax + 64);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->scratch_out);
  const void* src[4] = {xyz, time, laser_id, intensity};
  for (int i = 0; i < 4; ++i) LSA_HIP(ctx, hipMemcpyAsync(base + off[i], src[i], sz[i], hipMemcpyHostToDevice, ctx->stream));
  SoaFrame f;
  f.xyz = base + off[0]; f.time = base + off[1]; f.laser = base + off[2]; f.intensity = base + off[3];
  f.xyz_type = xyz_type; f.time_type = time_type; f.laser_type = laser_type; f.intensity_type = intensity_type;
  f.n = n;
  f.mapping_len = mapping_len;
  f.factor = time_to_seconds;
  if (mapping_len > 0) std::memcpy(f.mapping, laser_id_mapping, (size_t)mapping_len * sizeof(uint16_t));
  int* counts = reinterpret_cast<int*>(base + off_counts);
  long long* tmax = reinterpret_cast<long long*>(base + off_tmax);
  const long long lowest = (long long)0x8000000000000000ull;
  LSA_HIP(ctx, hipMemcpyAsync(tmax, &lowest, sizeof(lowest), hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps(ctx, "polydata_to_points", (double)total + (double)n * 32);
    hipLaunchKernelGGL(k_soa_scan_chunks, dim3(nchunks), dim3(256), 0, ctx->stream, f, tmax, counts);
    hipLaunchKernelGGL(k_soa_scan_counts, dim3(1), dim3(1024), 0, ctx->stream, counts, nchunks, counts + nchunks);
    hipLaunchKernelGGL(k_soa_to_points, dim3(nchunks), dim3(256), 0, ctx->stream, f, tmax, counts, reinterpret_cast<float4*>(ctx->frame_own));
  }
  long long tb = 0;
  int kept = 0;
  LSA_HIP(ctx, hipMemcpyAsync(&tb, tmax, sizeof(tb), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipMemcpyAsync(&kept, counts + nchunks, sizeof(kept), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's arrays may go away; stamp and size are needed now
  tb = tb >= 0 ? tb : tb ^ 0x7fffffffffffffffll;
  double end_time;
  std::memcpy(&end_time, &tb, sizeof(end_time));
  if (stamp_us) *stamp_us = (uint64_t)(end_time * (time_to_seconds * 1e6));  // pc->header.stamp = frameEndTime * (factor * 1e6) (:683)
  if (n_valid) *n_valid = kept;
  ctx->frame = kept > 0 ? ctx->frame_own : nullptr;
  ctx->frame_n = kept;
  ctx->inbox_current = -1;
  if (kept > 0 && (ctx->az_res < 1e-6 || M_PI / 4. < ctx->az_res))
  {
    // first usable frame: the azimuthal resolution is estimated on the host from the converted points (SSKE.cxx:593-637)
    std::vector<lsa_point_t> pts(kept);
    LSA_HIP(ctx, hipMemcpy(pts.data(), ctx->frame_own, (size_t)kept * sizeof(lsa_point_t), hipMemcpyDeviceToHost));
    maybe_estimate_resolution(ctx, pts.data(), kept);
  }
  return kept == n ? 1 : 0;  // allPointsAreValid
}

// ---- RobosenseToLidarNode::Callback (ros_wrapping/lidar_conversions/src/RobosenseToLidarNode.cxx:58-125) on the device ----
namespace
{
struct RsFrame
{
  const unsigned char* raw;
  int step, off_x, off_y, off_z, off_i;
  int n, width, points_per_ring, nlasers;
  int mapping_len, device_id;
  double rpm;
  uint16_t mapping[kMaxRings];
};
__device__ __forceinline__ float rs_f32(const RsFrame& f, int i, int off) { return *reinterpret_cast<const float*>(f.raw + (size_t)i * f.step + off); }
__device__ __forceinline__ bool rs_finite(const RsFrame& f, int i) { return isfinite(rs_f32(f, i, f.off_x)) && isfinite(rs_f32(f, i, f.off_y)) && isfinite(rs_f32(f, i, f.off_z)); }
// the last record with finite coordinates of every chunk of 1024 (-1: none)
__global__ __launch_bounds__(256) void k_rs_last_finite(RsFrame f, int* __restrict__ chunk_last)
{
  __shared__ int last;
  if (threadIdx.x == 0) last = -1;
  __syncthreads();
  int mine = -1;
  for (int q = 0; q < 4; ++q)
  {
    const int i = blockIdx.x * 1024 + q * 256 + threadIdx.x;
    if (i < f.n && rs_finite(f, i)) mine = i;
  }
  if (mine >= 0) atomicMax(&last, mine);
  __syncthreads();
  if (threadIdx.x == 0) chunk_last[blockIdx.x] = last;
}
// chunk_last -> the last finite record IN FRONT of every chunk (exclusive running maximum; one block)
__global__ __launch_bounds__(1024) void k_rs_carry(int* __restrict__ chunk_last, int nchunks)
{
  __shared__ int s[1024];
  int run = -1;
  for (int base = 0; base < nchunks; base += 1024)
  {
    const int i = base + threadIdx.x;
    const int v = i < nchunks ? chunk_last[i] : -1;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1)
    {
      const int a = threadIdx.x >= (unsigned)o ? s[threadIdx.x - o] : -1;
      __syncthreads();
      s[threadIdx.x] = max(s[threadIdx.x], a);
      __syncthreads();
    }
    const int before = threadIdx.x > 0 ? s[threadIdx.x - 1] : -1;
    if (i < nchunks) chunk_last[i] = max(run, before);
    run = max(run, s[1023]);
    __syncthreads();
  }
}
// A record stays when its coordinates are finite and differ from those of the last point KEPT.  A record skipped as a
// duplicate has the coordinates of the point kept before it, so "the last point kept" and "the nearest finite record in
// front" have the same coordinates: no sequential pass is needed.  keep[i], and how many stay per chunk.
__global__ __launch_bounds__(256) void k_rs_keep(RsFrame f, const int* __restrict__ chunk_carry, uint8_t* __restrict__ keep, int* __restrict__ chunk_count)
{
  __shared__ uint8_t fin[1024];
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  const int c0 = blockIdx.x * 1024;
  for (int q = 0; q < 4; ++q)
  {
    const int l = q * 256 + threadIdx.x, i = c0 + l;
    fin[l] = (i < f.n && rs_finite(f, i)) ? 1 : 0;
  }
  __syncthreads();
  int mine = 0;
  for (int q = 0; q < 4; ++q)
  {
    const int l = q * 256 + threadIdx.x, i = c0 + l;
    if (i >= f.n) continue;
    bool k = false;
    if (fin[l])
    {
      int j = l - 1;
      while (j >= 0 && !fin[j]) --j;
      const int prev = j >= 0 ? c0 + j : chunk_carry[blockIdx.x];
      k = prev < 0 || !(rs_f32(f, i, f.off_x) == rs_f32(f, prev, f.off_x) && rs_f32(f, i, f.off_y) == rs_f32(f, prev, f.off_y) &&
                        rs_f32(f, i, f.off_z) == rs_f32(f, prev, f.off_z));
    }
    keep[i] = k ? 1 : 0;
    mine += k ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) chunk_count[blockIdx.x] = cnt;
}
// the points that stay, in order, as LidarPoints
__global__ __launch_bounds__(256) void k_rs_to_points(RsFrame f, const uint8_t* __restrict__ keep, const int* __restrict__ chunk_start, float4* __restrict__ out)
{
  __shared__ int wave_base[4];
  int run = chunk_start[blockIdx.x];
  for (int q = 0; q < 4; ++q)
  {
    const int i = blockIdx.x * 1024 + q * 256 + threadIdx.x;
    const bool k = i < f.n && keep[i];
    const unsigned long long ballot = __ballot(k);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wave_base[wv] = __popcll(ballot);
    __syncthreads();
    int base = run;
    for (int w = 0; w < wv; ++w) base += wave_base[w];
    const int batch = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
    if (k)
    {
      const int at = base + __popcll(ballot & ((1ull << lane) - 1ull));
      const unsigned ring = (unsigned)i / (unsigned)f.width;
      // LaserIdMapping if given, RS16's when the input has 16 rings, otherwise the ring itself (:106-109)
      const unsigned rs16 = ring < 8u ? ring : 23u - ring;  // {0..7, 15, 14, ..., 8}
      const unsigned id = f.mapping_len > 0 ? (ring < (unsigned)f.mapping_len ? f.mapping[ring] : 0xffffu) : (f.nlasers == 16 ? rs16 : ring);
      const double adv = (double)((unsigned)i % (unsigned)f.points_per_ring) / (double)f.points_per_ring;
      const double t = (adv - 1) / f.rpm * 60.;
      const long long bits = __double_as_longlong(t);
      float4 a = make_float4(rs_f32(f, i, f.off_x), rs_f32(f, i, f.off_y), rs_f32(f, i, f.off_z), 1.f), b;
      b.x = __int_as_float((int)(bits & 0xffffffffll));
      b.y = __int_as_float((int)(bits >> 32));
      b.z = rs_f32(f, i, f.off_i);
      b.w = __uint_as_float((id & 0xffffu) | ((unsigned)(f.device_id & 0xff) << 16));  // laser_id, device_id, label 0
      out[2 * (size_t)at] = a;
      out[2 * (size_t)at + 1] = b;
    }
    run += batch;
    __syncthreads();
  }
}
}  // namespace

int lsa_upload_robosense_frame(lsa_ctx* ctx, const void* records, int width, int height, const lsa_wire_layout_t* lay, const uint16_t* laser_id_mapping,
                               int mapping_len, int device_id, double rpm, int* n_valid)
{
  if (!ctx || !records || !lay || width <= 0 || height <= 0 || lay->point_step <= 0 || mapping_len < 0 || (mapping_len > 0 && !laser_id_mapping) || !(rpm > 0.) ||
      (long long)width * height > (1ll << 30))
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_upload_robosense_frame: empty frame or bad layout") : LSA_E_ARG;
  if (mapping_len > kMaxRings) return ctx->fail(LSA_E_CAPACITY, "lsa_upload_robosense_frame: more than 512 entries in the laser id mapping");
  if (mapping_len > 0 && mapping_len < height) return ctx->fail(LSA_E_ARG, "lsa_upload_robosense_frame: the laser id mapping is shorter than the cloud is high");
  const int offs[4] = {lay->off_x, lay->off_y, lay->off_z, lay->off_intensity};
  for (int o : offs)
    if (o < 0 || o + 4 > lay->point_step || (o & 3)) return ctx->fail(LSA_E_ARG, "lsa_upload_robosense_frame: field outside the record or not aligned");
  if (lay->point_step & 3) return ctx->fail(LSA_E_ARG, "lsa_upload_robosense_frame: records must be a multiple of 4 bytes");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  const int n = width * height;
  int rc = ensure_capacity(ctx, n);
  if (rc) return rc;
  const size_t bytes = (size_t)n * lay->point_step;
  const int nchunks = (n + 1023) / 1024;
  const size_t off_keep = (bytes + 255) / 256 * 256, off_last = off_keep + ((size_t)n + 255) / 256 * 256,
               off_counts = off_last + ((size_t)nchunks * sizeof(int) + 255) / 256 * 256;
  rc = ensure_scratch(ctx, off_counts + (size_t)(nchunks + 1) * sizeof(int) + 64);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->scratch_out);
  LSA_HIP(ctx, hipMemcpyAsync(base, records, bytes, hipMemcpyHostToDevice, ctx->stream));
  RsFrame f;
  f.raw = reinterpret_cast<const unsigned char*>(base);
  f.step = lay->point_step; f.off_x = lay->off_x; f.off_y = lay->off_y; f.off_z = lay->off_z; f.off_i = lay->off_intensity;
  f.n = n; f.width = width; f.nlasers = height; f.points_per_ring = n / height;
  f.mapping_len = mapping_len; f.device_id = device_id; f.rpm = rpm;
  if (mapping_len > 0) std::memcpy(f.mapping, laser_id_mapping, (size_t)mapping_len * sizeof(uint16_t));
  uint8_t* keep = reinterpret_cast<uint8_t*>(base + off_keep);
  int* last = reinterpret_cast<int*>(base + off_last);
  int* counts = reinterpret_cast<int*>(base + off_counts);
  {
    ProfScope ps(ctx, "robosense_to_points", (double)bytes + (double)n * 32);
    hipLaunchKernelGGL(k_rs_last_finite, dim3(nchunks), dim3(256), 0, ctx->stream, f, last);
    hipLaunchKernelGGL(k_rs_carry, dim3(1), dim3(1024), 0, ctx->stream, last, nchunks);
    hipLaunchKernelGGL(k_rs_keep, dim3(nchunks), dim3(256), 0, ctx->stream, f, last, keep, counts);
    hipLaunchKernelGGL(k_soa_scan_counts, dim3(1), dim3(1024), 0, ctx->stream, counts, nchunks, counts + nchunks);
    hipLaunchKernelGGL(k_rs_to_points, dim3(nchunks), dim3(256), 0, ctx->stream, f, keep, counts, reinterpret_cast<float4*>(ctx->frame_own));
  }
  int kept = 0;
  LSA_HIP(ctx, hipMemcpyAsync(&kept, counts + nchunks, sizeof(kept), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's records may go away; the size is needed now
  if (n_valid) *n_valid = kept;
  ctx->frame = kept > 0 ? ctx->frame_own : nullptr;
  ctx->frame_n = kept;
  ctx->inbox_current = -1;
  if (kept > 0 && (ctx->az_res < 1e-6 || M_PI / 4. < ctx->az_res))
  {
    // first usable frame: the azimuthal resolution is estimated on the host from the converted points (SSKE.cxx:593-637)
    std::vector<lsa_point_t> pts(kept);
    LSA_HIP(ctx, hipMemcpy(pts.data(), ctx->frame_own, (size_t)kept * sizeof(lsa_point_t), hipMemcpyDeviceToHost));
    maybe_estimate_resolution(ctx, pts.data(), kept);
  }
  return LSA_OK;
}

int lsa_frame_store_put(lsa_ctx* ctx, int slot, const lsa_point_t* pts, int n)
{
  if (!ctx || !pts || n <= 0 || slot < 0 || slot > 65536) return ctx ? ctx->fail(LSA_E_ARG, "lsa_frame_store_put: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_capacity(ctx, n);
  if (rc) return rc;
  if ((int)ctx->store.size() <= slot) { ctx->store.resize(slot + 1, {nullptr, 0}); ctx->store_cap.resize(slot + 1, 0); }
  lsa_point_t* d = ctx->store[slot].first;
  if (ctx->prefetch_pending && d && ctx->prefetch_frame == d)
  {
    // the look-ahead extraction reads this slot: let it finish, its result no longer describes the slot
    LSA_HIP(ctx, hipStreamSynchronize(ctx->prefetch_stream));
    ctx->prefetch_pending = false;
  }
  if (!d || ctx->store_cap[slot] < n)
  {
    // the frame in use may be this very slot: nothing may still read it
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (d) { if (ctx->frame == d) { ctx->frame = nullptr; ctx->frame_n = 0; } retire_dev(ctx, d); ctx->store[slot] = {nullptr, 0}; ctx->store_cap[slot] = 0; }
    d = nullptr;
    LSA_HIP(ctx, hipMalloc((void**)&d, (size_t)n * sizeof(lsa_point_t)));
    ctx->store_cap[slot] = n;
  }
  LSA_HIP(ctx, hipMemcpyAsync(d, pts, (size_t)n * sizeof(lsa_point_t), hipMemcpyHostToDevice, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pts may be pageable and reused by the caller
  ctx->store[slot] = {d, n};
  if (ctx->frame == d) ctx->frame_n = n;
  maybe_estimate_resolution(ctx, pts, n);
  return LSA_OK;
}

int lsa_frame_store_use(lsa_ctx* ctx, int slot)
{
  if (!ctx || slot < 0 || slot >= (int)ctx->store.size() || !ctx->store[slot].first)
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_frame_store_use: empty slot") : LSA_E_ARG;
  ctx->frame = ctx->store[slot].first;
  ctx->frame_n = ctx->store[slot].second;
  ctx->inbox_current = -1;
  return LSA_OK;
}

// The host threads of a context talk to the device in tens of short round trips per frame (mailbox polls, pinned
// staging buffers): on a two-socket host they belong on the socket the GPU hangs off.
int lsa_bind_host_to_device(int device_id)
{
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), device_id) != hipSuccess) return LSA_E_NO_DEVICE;
  for (char* c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
  int node = -1;
  {
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    if (FILE* f = std::fopen(path.c_str(), "r")) { if (std::fscanf(f, "%d", &node) != 1) node = -1; std::fclose(f); }
  }
  if (node < 0) return LSA_E_STATE;  // single-socket host, or the kernel does not say
  cpu_set_t want;
  CPU_ZERO(&want);
  {
    const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return LSA_E_STATE;
    int a = 0, b = 0;
    // "0-63,128-191"
    while (std::fscanf(f, "%d", &a) == 1)
    {
      b = a;
      int ch = std::fgetc(f);
      if (ch == '-') { if (std::fscanf(f, "%d", &b) != 1) b = a; ch = std::fgetc(f); }
      for (int c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET(c, &want);
      if (ch != ',') break;
    }
    std::fclose(f);
  }
  cpu_set_t have;
  CPU_ZERO(&have);
  if (sched_getaffinity(0, sizeof(have), &have) != 0) return LSA_E_STATE;
  cpu_set_t both;
  CPU_AND(&both, &want, &have);
  if (CPU_COUNT(&both) == 0) return LSA_E_STATE;  // the node's CPUs are not ours to use
  if (sched_setaffinity(0, sizeof(both), &both) != 0) return LSA_E_STATE;
  return node;
}

int lsa_frame_size(const lsa_ctx* ctx) { return ctx ? ctx->frame_n : 0; }
float lsa_get_azimuthal_resolution(const lsa_ctx* ctx) { return ctx ? ctx->az_res : 0.f; }
void lsa_set_azimuthal_resolution(lsa_ctx* ctx, float rad) { if (ctx) ctx->az_res = rad; }
int lsa_nb_laser_rings(const lsa_ctx* ctx) { return ctx ? ctx->nb_rings_seen : 0; }

// what two events measure with nothing between them (the markers' own way through the queue): the median of 15 pairs on the
// idle stream.  A scope's time is what its events measure minus this, so that it can be held against a profiler's figure
// for the kernel alone.
static void calibrate_event_overhead(lsa_ctx* ctx)
{
  if (ctx->prof_overhead_ms > 0.) return;
  hipEvent_t a = nullptr, b = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  (void)hipStreamSynchronize(ctx->stream);
  std::vector<float> ms;
  for (int i = 0; i < 15; ++i)
  {
    (void)hipEventRecord(a, ctx->stream);
    (void)hipEventRecord(b, ctx->stream);
    (void)hipEventSynchronize(b);
    float t = 0;
    if (hipEventElapsedTime(&t, a, b) == hipSuccess) ms.push_back(t);
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  if (ms.empty()) return;
  std::sort(ms.begin(), ms.end());
  ctx->prof_overhead_ms = ms[ms.size() / 2];
}

double lsa_profile_event_overhead_us(const lsa_ctx* ctx) { return ctx ? 1e3 * ctx->prof_overhead_ms : 0.; }

int lsa_profile_enable(lsa_ctx* ctx, int on)
{
  if (!ctx) return LSA_E_ARG;
  if (on) calibrate_event_overhead(ctx);
  ctx->profiling = on != 0;
  ctx->prof_only.clear();
  ctx->prof_every = 1;
  return LSA_OK;
}
int lsa_profile_select(lsa_ctx* ctx, const char* scope, int every)
{
  if (!ctx || !scope || every < 1) return LSA_E_ARG;
  calibrate_event_overhead(ctx);
  ctx->profiling = true;
  ctx->prof_only = scope;
  ctx->prof_every = every;
  return LSA_OK;
}
int lsa_profile_reset(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  profile_collect(ctx);
  ctx->stats.clear();
  return LSA_OK;
}
int lsa_profile_get(lsa_ctx* ctx, lsa_kernel_stat_t* out, int capacity)
{
  if (!ctx) return LSA_E_ARG;
  (void)hipStreamSynchronize(ctx->stream);
  profile_collect(ctx);
  int n = std::min<int>(capacity, ctx->stats.size());
  for (int i = 0; i < n; ++i)
  {
    std::memset(&out[i], 0, sizeof(out[i]));
    std::strncpy(out[i].name, ctx->stats[i].name.c_str(), sizeof(out[i].name) - 1);
    out[i].launches = ctx->stats[i].launches;
    // sampled scopes: the timed launches' mean stands for all launches
    out[i].total_ms = ctx->stats[i].timed > 0 ? ctx->stats[i].total_ms * ((double)ctx->stats[i].launches / ctx->stats[i].timed) : 0.;
    out[i].bytes = ctx->stats[i].bytes;
  }
  return n;
}

}  // extern "C"
