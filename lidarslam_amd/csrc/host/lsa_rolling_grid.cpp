// lsa_rolling_grid.cpp -- see lsa_rolling_grid.h.  Float/double mixing follows Eigen's scalar
// promotion in the reference (an Array3f combined with a double scalar casts the scalar to float).
#include "lsa_rolling_grid.h"
#include <algorithm>
#include <cmath>
#include <limits>

namespace lsa
{
namespace host
{
namespace
{
// Utils::PositionToVoxel (RollingGrid.h:39-42): round((p - origin) / resolution)
// static_cast<int>(std::round(x)) without the libm call: truncate, then step away from zero when the
// remainder reaches one half (x - float(t) is exact for |x| < 2^24, far above any voxel index)
// Out-of-range values saturate (the reference's cast is undefined there): the bounding box of an empty
// keypoint set, (+FLT_MAX, -FLT_MAX), then selects no voxel, as it does in the reference in practice.
inline int RoundToInt(float x)
{
  if (!(x > -1.0e9f)) return -1000000000;
  if (!(x < 1.0e9f)) return 1000000000;
  int t = static_cast<int>(x);
  const float d = x - static_cast<float>(t);
  if (d >= 0.5f) ++t;
  else if (d <= -0.5f) --t;
  return t;
}
inline void ToVoxel(const float p[3], const float origin[3], double resolution, int out[3])
{
  const float r = static_cast<float>(resolution);
  for (int i = 0; i < 3; ++i) out[i] = RoundToInt((p[i] - origin[i]) / r);
}
inline void BoundingBox(const lsa_point_t* pts, size_t count, float mn[3], float mx[3])
{
  for (int i = 0; i < 3; ++i)
  {
    mn[i] = std::numeric_limits<float>::max();
    mx[i] = -std::numeric_limits<float>::max();
  }
  for (size_t i = 0; i < count; ++i)
  {
    const lsa_point_t& p = pts[i];
    mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
    mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
    mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
  }
}
inline float Dist(const lsa_point_t& p, const float c[3])
{
  const float dx = p.x - c[0], dy = p.y - c[1], dz = p.z - c[2];
  return std::sqrt(dx * dx + (dy * dy + dz * dz));
}
}  // namespace

void RollingGrid::To3d(int id, int v[3]) const
{
  const int z = id / (GridSize * GridSize);
  id -= z * GridSize * GridSize;
  const int y = id / GridSize;
  id -= y * GridSize;
  v[0] = id; v[1] = y; v[2] = z;
}

void RollingGrid::GridOrigin(float o[3]) const
{
  // RollingGrid.cxx:177: centre of voxel (0,0,0)
  for (int i = 0; i < 3; ++i) o[i] = VoxelGridPosition[i] - static_cast<float>(int(GridSize / 2) * VoxelResolution);
}

// RollingGrid.cxx:40-48
void RollingGrid::Reset(const float position[3])
{
  this->Clear();
  const float r = static_cast<float>(VoxelResolution);
  for (int i = 0; i < 3; ++i) VoxelGridPosition[i] = std::floor((position ? position[i] : 0.f) / r) * r;
}

// RollingGrid.cxx:51-56
void RollingGrid::Clear()
{
  NbPoints = 0;
  Voxels.clear();
  SubMapValid = false;
}

// RollingGrid.cxx:59-70
void RollingGrid::SetGridSize(int size)
{
  GridSize = size;
  PointCloud prev = this->Get();
  this->Clear();
  if (!prev.empty()) this->Add(prev);
}

// RollingGrid.cxx:73-88
void RollingGrid::SetVoxelResolution(double resolution)
{
  VoxelResolution = int(resolution / LeafSize) * LeafSize;
  const float r = static_cast<float>(VoxelResolution);
  for (int i = 0; i < 3; ++i) VoxelGridPosition[i] = std::floor(VoxelGridPosition[i] / r) * r;
  PointCloud prev = this->Get();
  this->Clear();
  if (!prev.empty()) this->Add(prev);
}

// RollingGrid.cxx:95-114
template <typename Pred> void RollingGrid::AppendOrdered(Pred pred, lsa_point_t* out, std::size_t& count) const
{
  struct Ref { unsigned out, in; const Voxel* v; };
  std::vector<Ref> refs;
  refs.reserve(NbPoints);
  for (const auto& o : Voxels)
    for (const auto& i : o.second)
      if (pred(o.first, i.second)) refs.push_back({static_cast<unsigned>(o.first), static_cast<unsigned>(i.first), &i.second});
  std::sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.out != b.out ? a.out < b.out : a.in < b.in; });
  for (const Ref& r : refs) out[count++] = r.v->point;
}

RollingGrid::PointCloud RollingGrid::Get(bool clean) const
{
  PointCloud pc;
  if (Ordered)
  {
    pc.resize(NbPoints);
    std::size_t n = 0;
    const unsigned minFrames = MinFramesPerVoxel;
    this->AppendOrdered([&](int, const Voxel& v) { return !clean || v.count > minFrames; }, pc.data(), n);
    pc.resize(n);
    return pc;
  }
  pc.reserve(NbPoints);
  for (const auto& out : Voxels)
    for (const auto& in : out.second)
      if (!clean || in.second.count > MinFramesPerVoxel) pc.push_back(in.second.point);
  return pc;
}

// RollingGrid.cxx:117-157
void RollingGrid::Roll(const float minPoint[3], const float maxPoint[3])
{
  const float half = static_cast<float>(static_cast<double>(GridSize) / 2 * VoxelResolution);
  const float res = static_cast<float>(VoxelResolution);
  int shift[3];
  bool move = false;
  for (int i = 0; i < 3; ++i)
  {
    const float down = minPoint[i] - (VoxelGridPosition[i] - half);
    const float up = maxPoint[i] - (VoxelGridPosition[i] + half);
    float off = (up + down) / 2.f;
    off = std::min(std::max(off, std::min(down, 0.f)), std::max(up, 0.f));
    shift[i] = static_cast<int>(std::round(off / res));
    move = move || shift[i] != 0;
  }
  if (!move) return;
  unsigned int kept = 0;
  RollingVG rolled;
  for (auto& out : Voxels)
  {
    int v[3];
    this->To3d(out.first, v);
    bool inside = true;
    for (int i = 0; i < 3; ++i)
    {
      v[i] -= shift[i];
      inside = inside && 0 <= v[i] && v[i] < GridSize;
    }
    if (inside)
    {
      kept += out.second.size();
      rolled[this->To1d(v)] = std::move(out.second);
    }
  }
  NbPoints = kept;
  Voxels.swap(rolled);
  for (int i = 0; i < 3; ++i) VoxelGridPosition[i] += static_cast<float>(shift[i]) * res;
}

// RollingGrid.cxx:160-318
void RollingGrid::Add(const lsa_point_t* points, size_t count, bool fixed, double currentTime, bool roll)
{
  if (count == 0) return;
  if (roll)
  {
    float mn[3], mx[3];
    BoundingBox(points, count, mn, mx);
    this->Roll(mn, mx);
  }
  const float res = static_cast<float>(VoxelResolution);
  const float leaf = static_cast<float>(LeafSize);
  float origin[3];
  this->GridOrigin(origin);

  // The reference tracks "first time this leaf voxel is touched by this call" in a map of maps
  // (`seen`, RollingGrid.cxx:183) and reaches the voxel through Voxels[idxOut][idxIn] four times per
  // point; here the voxel remembers the serial of the last Add that touched it and is looked up once.
  // Insertions happen at the same moments (outer voxel on first use, then the leaf), so the iteration
  // order of both maps -- and with it the order of the sub-map points -- is unchanged.
  if (AddCrew && Sampling != SamplingMode::CENTROID && count >= 2048)
  {
    this->AddParallel(points, count, fixed, currentTime);
    return;
  }
  const unsigned int serial = ++AddSerial;
  std::unordered_map<int, std::unordered_map<int, Voxel>> meanPts;
  bool updated = false;
  int lastOut = -1;
  SamplingVG* outer = nullptr;
  for (size_t pi = 0; pi < count; ++pi)
  {
    const lsa_point_t& point = points[pi];
    const float p[3] = {point.x, point.y, point.z};
    int vo[3];
    ToVoxel(p, origin, VoxelResolution, vo);
    if (!(0 <= vo[0] && vo[0] < GridSize && 0 <= vo[1] && vo[1] < GridSize && 0 <= vo[2] && vo[2] < GridSize)) continue;
    float centerIn[3];
    for (int i = 0; i < 3; ++i) centerIn[i] = static_cast<float>(vo[i]) * res + origin[i];
    int vi[3];
    ToVoxel(p, centerIn, LeafSize, vi);
    const int idxOut = this->To1d(vo);
    const int idxIn = this->To1d(vi);
    if (idxOut != lastOut || !outer)
    {
      outer = &Voxels[idxOut];  // inserts the outer voxel when it is new, exactly as Voxels[idxOut][idxIn] would
      lastOut = idxOut;
    }
    auto ins = outer->try_emplace(idxIn);  // no node is built (and thrown away) when the voxel exists
    Voxel& voxel = ins.first->second;
    if (ins.second)
    {
      voxel.point = point;
      ++NbPoints;
      updated = true;
    }
    else
    {
      if (voxel.point.label == 1) continue;  // fixed map point
      switch (Sampling)
      {
        case SamplingMode::FIRST:
          break;
        case SamplingMode::LAST:
          voxel.point = point;
          updated = true;
          break;
        case SamplingMode::MAX_INTENSITY:
          if (point.intensity > voxel.point.intensity)
          {
            voxel.point = point;
            updated = true;
          }
          break;
        case SamplingMode::CENTER_POINT:
        {
          float c[3];
          for (int i = 0; i < 3; ++i) c[i] = centerIn[i] - static_cast<float>(VoxelResolution / 2.f) + leaf * static_cast<float>(vi[i]);
          if (Dist(point, c) < Dist(voxel.point, c))
          {
            voxel.point = point;
            updated = true;
          }
          break;
        }
        case SamplingMode::CENTROID:
        {
          Voxel& v = meanPts[idxOut][idxIn];
          const float k = static_cast<float>(v.count), k1 = static_cast<float>(v.count + 1);
          v.point.x = (v.point.x * k + point.x) / k1;
          v.point.y = (v.point.y * k + point.y) / k1;
          v.point.z = (v.point.z * k + point.z) / k1;
          ++v.count;
          break;
        }
      }
    }
    if (Sampling == SamplingMode::CENTROID)
    {
      // the reference runs this block for every added point (RollingGrid.cxx:282-297)
      for (auto& mo : meanPts)
        for (auto& mi : mo.second)
        {
          Voxel& vx = Voxels[mo.first][mi.first];
          const float k = static_cast<float>(vx.count), k1 = static_cast<float>(vx.count + 1);
          vx.point.x = (vx.point.x * k + mi.second.point.x) / k1;
          vx.point.y = (vx.point.y * k + mi.second.point.y) / k1;
          vx.point.z = (vx.point.z * k + mi.second.point.z) / k1;
        }
    }
    voxel.point.time = currentTime;
    voxel.point.label = fixed ? 1 : 0;
    if (voxel.seen != serial)
    {
      ++voxel.count;
      voxel.seen = serial;
    }
  }
  if (updated) SubMapValid = false;  // KdTree.Reset() in the reference
}

// The per-point body of Add() for every sampling mode but CENTROID (whose running means couple the voxels).
inline void RollingGrid::AddOne(const lsa_point_t& point, SamplingVG& outer, int idxIn, const int vi[3], const float centerIn[3], bool fixed,
                                double currentTime, unsigned int serial, AddTally& tally)
{
  auto ins = outer.try_emplace(idxIn);
  Voxel& voxel = ins.first->second;
  if (ins.second)
  {
    voxel.point = point;
    ++tally.inserted;
    tally.updated = true;
  }
  else
  {
    if (voxel.point.label == 1) return;  // fixed map point
    switch (Sampling)
    {
      case SamplingMode::LAST:
        voxel.point = point;
        tally.updated = true;
        break;
      case SamplingMode::MAX_INTENSITY:
        if (point.intensity > voxel.point.intensity)
        {
          voxel.point = point;
          tally.updated = true;
        }
        break;
      case SamplingMode::CENTER_POINT:
      {
        const float leaf = static_cast<float>(LeafSize);
        float c[3];
        for (int i = 0; i < 3; ++i) c[i] = centerIn[i] - static_cast<float>(VoxelResolution / 2.f) + leaf * static_cast<float>(vi[i]);
        if (Dist(point, c) < Dist(voxel.point, c))
        {
          voxel.point = point;
          tally.updated = true;
        }
        break;
      }
      default:
        break;  // FIRST
    }
  }
  voxel.point.time = currentTime;
  voxel.point.label = fixed ? 1 : 0;
  if (voxel.seen != serial)
  {
    ++voxel.count;
    voxel.seen = serial;
  }
}

// Add() on several threads, same maps in the end (content and iteration order): see SetAddThreads.
void RollingGrid::AddParallel(const lsa_point_t* points, std::size_t count, bool fixed, double currentTime)
{
  const float res = static_cast<float>(VoxelResolution);
  float origin[3];
  this->GridOrigin(origin);
  const unsigned int serial = ++AddSerial;
  const int threads = AddCrew->Size();
  AddOut.resize(count);
  AddIn.resize(count);
  // 1. voxel indices of every point (pure arithmetic, contiguous shares)
  AddCrew->Run([&](int tid) {
    const std::size_t lo = count * tid / threads, hi = count * (tid + 1) / threads;
    for (std::size_t pi = lo; pi < hi; ++pi)
    {
      const float p[3] = {points[pi].x, points[pi].y, points[pi].z};
      int vo[3];
      ToVoxel(p, origin, VoxelResolution, vo);
      if (!(0 <= vo[0] && vo[0] < GridSize && 0 <= vo[1] && vo[1] < GridSize && 0 <= vo[2] && vo[2] < GridSize)) { AddOut[pi] = -1; continue; }
      float centerIn[3];
      for (int i = 0; i < 3; ++i) centerIn[i] = static_cast<float>(vo[i]) * res + origin[i];
      int vi[3];
      ToVoxel(p, centerIn, LeafSize, vi);
      AddOut[pi] = this->To1d(vo);
      AddIn[pi] = this->To1d(vi);
    }
  });
  // 2. the outer voxels, created in the order in which the points reach them (this thread alone), and how many
  //    points each receives: the keypoints crowd around the sensor, so the outer voxels are dealt out by weight
  std::unordered_map<int, int> load;  // outer voxel -> points, then -> owning thread
  {
    int lastOut = -1;
    int* counter = nullptr;
    for (std::size_t pi = 0; pi < count; ++pi)
    {
      const int idxOut = AddOut[pi];
      if (idxOut < 0) continue;
      if (idxOut != lastOut)
      {
        (void)Voxels[idxOut];
        counter = &load[idxOut];
        lastOut = idxOut;
      }
      ++*counter;
    }
    std::vector<std::pair<int, int>> byWeight;  // (points, outer voxel), heaviest first; ties by index: deterministic
    byWeight.reserve(load.size());
    for (const auto& kv : load) byWeight.emplace_back(kv.second, kv.first);
    std::sort(byWeight.begin(), byWeight.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) {
      return a.first != b.first ? a.first > b.first : a.second < b.second;
    });
    std::vector<long> assigned(threads, 0);
    for (const auto& w : byWeight)
    {
      const int t = static_cast<int>(std::min_element(assigned.begin(), assigned.end()) - assigned.begin());
      assigned[t] += w.first;
      load[w.second] = t;
    }
  }
  // 3. the leaf voxels: every outer voxel belongs to one thread, which takes its points in their order; the outer
  //    map is only read from here on
  std::vector<AddTally> tally(threads);
  AddCrew->Run([&](int tid) {
    AddTally mine;
    int lastOut = -1;
    bool mineNow = false;
    SamplingVG* outer = nullptr;
    for (std::size_t pi = 0; pi < count; ++pi)
    {
      const int idxOut = AddOut[pi];
      if (idxOut < 0) continue;
      if (idxOut != lastOut)
      {
        lastOut = idxOut;
        mineNow = load.find(idxOut)->second == tid;
        if (mineNow) outer = &Voxels.find(idxOut)->second;
      }
      if (!mineNow) continue;
      // the leaf coordinates and the outer voxel's centre again, for CENTER_POINT (cheap next to the hash lookup)
      int vo[3], vi[3];
      this->To3d(idxOut, vo);
      float centerIn[3];
      for (int i = 0; i < 3; ++i) centerIn[i] = static_cast<float>(vo[i]) * res + origin[i];
      if (Sampling == SamplingMode::CENTER_POINT)
      {
        const float p[3] = {points[pi].x, points[pi].y, points[pi].z};
        ToVoxel(p, centerIn, LeafSize, vi);
      }
      else vi[0] = vi[1] = vi[2] = 0;
      this->AddOne(points[pi], *outer, AddIn[pi], vi, centerIn, fixed, currentTime, serial, mine);
    }
    tally[tid] = mine;
  });
  bool updated = false;
  for (const AddTally& t : tally)
  {
    NbPoints += t.inserted;
    updated = updated || t.updated;
  }
  if (updated) SubMapValid = false;
}

// ---- Crew
Crew::Crew(int threads)
{
  for (int t = 1; t < threads; ++t) Helpers.emplace_back([this, t] { this->Loop(t); });
}

Crew::~Crew()
{
  Quit.store(true);
  {
    std::lock_guard<std::mutex> lock(M);
    Generation.fetch_add(1);
  }
  Cv.notify_all();
  for (auto& h : Helpers) h.join();
}

void Crew::Wake()
{
  if (Sleeping.load(std::memory_order_acquire) > 0)
  {
    { std::lock_guard<std::mutex> lock(M); }
    Cv.notify_all();
  }
}

void Crew::Loop(int tid)
{
  unsigned seen = 0;
  for (;;)
  {
    // spin for a while (about a quarter of a millisecond), then sleep until woken or until there is work
    int spins = 0;
    while (Generation.load(std::memory_order_acquire) == seen)
    {
      if (++spins < 20000) { __builtin_ia32_pause(); continue; }
      std::unique_lock<std::mutex> lock(M);
      if (Generation.load(std::memory_order_acquire) != seen) break;  // Run() bumps it under this lock
      Sleeping.fetch_add(1, std::memory_order_acq_rel);
      Cv.wait(lock);  // a wake-up without work just restarts the spinning
      Sleeping.fetch_sub(1, std::memory_order_acq_rel);
      spins = 0;
    }
    seen = Generation.load(std::memory_order_acquire);
    if (Quit.load()) return;
    (*Fn)(tid);
    Pending.fetch_sub(1, std::memory_order_acq_rel);
  }
}

void Crew::Run(const std::function<void(int)>& fn)
{
  Fn = &fn;
  Pending.store(static_cast<int>(Helpers.size()), std::memory_order_release);
  {
    // under the lock: a helper about to sleep either sees the new generation or is notified
    std::lock_guard<std::mutex> lock(M);
    Generation.fetch_add(1, std::memory_order_acq_rel);
  }
  if (Sleeping.load(std::memory_order_acquire) > 0) Cv.notify_all();
  fn(0);
  while (Pending.load(std::memory_order_acquire) > 0)
  {
    if (Sleeping.load(std::memory_order_acquire) > 0) Cv.notify_all();  // a helper that went to sleep at the last moment
    __builtin_ia32_pause();
  }
}

void RollingGrid::SetAddThreads(int n)
{
  n = std::max(1, std::min(n, 16));
  if (n == this->GetAddThreads()) return;
  AddCrew.reset();
  if (n > 1) AddCrew.reset(new Crew(n));
}

// RollingGrid.cxx:325-351
void RollingGrid::ClearOldPoints(double currentTime)
{
  for (auto out = Voxels.begin(); out != Voxels.end();)
  {
    for (auto in = out->second.begin(); in != out->second.end();)
    {
      const Voxel& voxel = in->second;
      if (!voxel.point.label && currentTime - voxel.point.time > DecayingThreshold) in = out->second.erase(in);
      else ++in;
    }
    if (out->second.empty()) out = Voxels.erase(out);
    else ++out;
  }
}

// RollingGrid.cxx:354-360
void RollingGrid::BeginSubMap(std::size_t capacity)
{
  SubMapCount = 0;
  if (SubMapStorage) SubMapPtr = SubMapStorage(capacity);
  else
  {
    SubMapOwn.resize(capacity);
    SubMapPtr = SubMapOwn.data();
  }
  if (!SubMapPtr && capacity > 0)
  {
    // the provider failed: fall back to own storage (the caller notices through SubMapData())
    SubMapOwn.resize(capacity);
    SubMapPtr = SubMapOwn.data();
  }
}

void RollingGrid::BuildSubMap()
{
  this->BeginSubMap(NbPoints);
  if (Ordered)
  {
    this->AppendOrdered([](int, const Voxel&) { return true; }, SubMapPtr, SubMapCount);
    SubMapValid = true;
    SubMapBoxed = false;
    return;
  }
  for (const auto& out : Voxels)
    for (const auto& in : out.second) SubMapPtr[SubMapCount++] = in.second.point;
  SubMapValid = true;
  SubMapBoxed = false;
}

void RollingGrid::VoxelRange(const float minPoint[3], const float maxPoint[3], int lo[3], int hi[3]) const
{
  float origin[3];
  this->GridOrigin(origin);
  ToVoxel(minPoint, origin, VoxelResolution, lo);
  ToVoxel(maxPoint, origin, VoxelResolution, hi);
  for (int i = 0; i < 3; ++i)
  {
    lo[i] = std::max(lo[i], 0);
    hi[i] = std::min(hi[i], GridSize - 1);
  }
}

bool RollingGrid::SubMapBuiltFor(const float minPoint[3], const float maxPoint[3], int minNbPoints) const
{
  if (!SubMapValid || !SubMapBoxed || minNbPoints != SubMapMinNbPoints) return false;
  int lo[3], hi[3];
  this->VoxelRange(minPoint, maxPoint, lo, hi);
  for (int i = 0; i < 3; ++i)
    if (lo[i] != SubMapLo[i] || hi[i] != SubMapHi[i]) return false;
  return true;
}

// RollingGrid.cxx:363-442
void RollingGrid::BuildSubMap(const float minPoint[3], const float maxPoint[3], int minNbPoints)
{
  int lo[3], hi[3];
  this->VoxelRange(minPoint, maxPoint, lo, hi);
  for (int i = 0; i < 3; ++i) { SubMapLo[i] = lo[i]; SubMapHi[i] = hi[i]; }
  SubMapMinNbPoints = minNbPoints;
  SubMapBoxed = true;
  auto intersects = [&](int id) {
    int v[3];
    this->To3d(id, v);
    return lo[0] <= v[0] && v[0] <= hi[0] && lo[1] <= v[1] && v[1] <= hi[1] && lo[2] <= v[2] && v[2] <= hi[2];
  };
  this->BeginSubMap(NbPoints);
  if (Ordered)
  {
    const unsigned minFrames = MinFramesPerVoxel;
    if (minNbPoints < 0 || MinFramesPerVoxel <= 1)
      this->AppendOrdered([&](int id, const Voxel&) { return intersects(id); }, SubMapPtr, SubMapCount);
    else
    {
      this->AppendOrdered([&](int id, const Voxel& v) { return intersects(id) && (v.count >= minFrames || v.point.label == 1); }, SubMapPtr, SubMapCount);
      if (static_cast<int>(SubMapCount) < minNbPoints)
        this->AppendOrdered([&](int id, const Voxel& v) { return intersects(id) && v.count < minFrames && v.point.label != 1; }, SubMapPtr, SubMapCount);
    }
    SubMapValid = true;
    return;
  }
  if ((minNbPoints < 0 || MinFramesPerVoxel <= 1) && AddCrew && NbPoints >= 8192)
  {
    // every outer voxel contributes all its points, as one run in the iteration order of the outer map: the runs'
    // places are known from the sizes alone, and the helper threads copy them side by side
    std::vector<const SamplingVG*> run;
    std::vector<std::size_t> at;
    std::size_t total = 0;
    for (const auto& out : Voxels)
      if (intersects(out.first))
      {
        run.push_back(&out.second);
        at.push_back(total);
        total += out.second.size();
      }
    std::atomic<std::size_t> next{0};
    lsa_point_t* dst = SubMapPtr;
    AddCrew->Run([&](int) {
      for (std::size_t i = next.fetch_add(1); i < run.size(); i = next.fetch_add(1))
      {
        lsa_point_t* p = dst + at[i];
        for (const auto& in : *run[i]) *p++ = in.second.point;
      }
    });
    SubMapCount = total;
  }
  else if (minNbPoints < 0 || MinFramesPerVoxel <= 1)
  {
    for (const auto& out : Voxels)
      if (intersects(out.first))
        for (const auto& in : out.second) SubMapPtr[SubMapCount++] = in.second.point;
  }
  else
  {
    for (const auto& out : Voxels)
      if (intersects(out.first))
        for (const auto& in : out.second)
          if (in.second.count >= MinFramesPerVoxel || in.second.point.label == 1) SubMapPtr[SubMapCount++] = in.second.point;
    if (static_cast<int>(SubMapCount) < minNbPoints)
    {
      for (const auto& out : Voxels)
        if (intersects(out.first))
          for (const auto& in : out.second)
            if (in.second.count < MinFramesPerVoxel && in.second.point.label != 1) SubMapPtr[SubMapCount++] = in.second.point;
    }
  }
  SubMapValid = true;
}

}  // namespace host
}  // namespace lsa
