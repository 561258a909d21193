// lsa_rolling_grid.h -- LidarSlam::RollingGrid on the host
// (slam_lib/include/LidarSlam/RollingGrid.h, slam_lib/src/RollingGrid.cxx).
//
// map maintenance stays on the host in this round (it runs once per keyframe
// and is hash-map bound); its OUTPUT, the sub-map cloud and its point order, is the kNN target
// that lsa_set_target uploads.  The containers are the reference's own
// std::unordered_map<int, std::unordered_map<int, Voxel>> so that libstdc++'s iteration order --
// which decides the order of the sub-map points and with it the PCA summation order -- is kept.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../../include/lidarslam_amd.h"

namespace lsa
{
namespace host
{

enum class SamplingMode { FIRST = 0, LAST = 1, MAX_INTENSITY = 2, CENTER_POINT = 3, CENTROID = 4 };

// A few helper threads that run one function together with the caller (thread 0) and meet again at the end.
// Between calls they spin for a short while (the phases of one Add follow each other within microseconds) and then
// sleep on a condition variable; Wake() ends the sleep ahead of the next call.
class Crew
{
public:
  explicit Crew(int threads);
  ~Crew();
  int Size() const { return static_cast<int>(Helpers.size()) + 1; }
  void Wake();
  void Run(const std::function<void(int)>& fn);

private:
  void Loop(int tid);
  std::vector<std::thread> Helpers;
  std::mutex M;
  std::condition_variable Cv;
  const std::function<void(int)>* Fn = nullptr;
  std::atomic<unsigned> Generation{0};
  std::atomic<int> Pending{0};
  std::atomic<bool> Quit{false};
  std::atomic<int> Sleeping{0};
};

class RollingGrid
{
public:
  using PointCloud = std::vector<lsa_point_t>;
  struct Voxel
  {
    lsa_point_t point{};
    unsigned int count = 0;
    unsigned int seen = 0;  // serial of the last Add() that touched this voxel (replaces the reference's `seen` maps)
  };
  using SamplingVG = std::unordered_map<int, Voxel>;
  using RollingVG = std::unordered_map<int, SamplingVG>;

  RollingGrid() { this->Reset(); }
  void Reset(const float position[3] = nullptr);
  void Clear();

  void SetGridSize(int size);
  int GetGridSize() const { return this->GridSize; }
  void SetVoxelResolution(double resolution);
  double GetVoxelResolution() const { return this->VoxelResolution; }
  void SetLeafSize(double s) { this->LeafSize = s; }
  double GetLeafSize() const { return this->LeafSize; }
  void SetMinFramesPerVoxel(unsigned int n) { this->MinFramesPerVoxel = n; }
  unsigned int GetMinFramesPerVoxel() const { return this->MinFramesPerVoxel; }
  void SetSampling(SamplingMode m) { this->Sampling = m; }
  SamplingMode GetSampling() const { return this->Sampling; }
  void SetDecayingThreshold(double d) { this->DecayingThreshold = d; }
  // Order in which Get / BuildSubMap hand the voxels out: true (default) ascending (outer index, leaf index as
  // unsigned), the defined order the device map uses; false: the iteration order of the reference's own containers.
  void SetOrdered(bool b) { this->Ordered = b; }
  bool GetOrdered() const { return this->Ordered; }
  // Threads Add() uses for a big cloud (1 = the caller alone).  The leaf voxels of different outer voxels are
  // independent containers: the outer voxels are created first, in the order of the points, then every thread
  // inserts the points of its share of the outer voxels, in the order of the points -- the maps end up with the
  // same content and the same iteration order as after the sequential loop.
  void SetAddThreads(int n);
  int GetAddThreads() const { return this->AddCrew ? this->AddCrew->Size() : 1; }
  void WakeAddThreads() { if (this->AddCrew) this->AddCrew->Wake(); }
  double GetDecayingThreshold() const { return this->DecayingThreshold; }
  bool IsTimeThreshold() const { return this->DecayingThreshold > 0; }

  PointCloud Get(bool clean = false) const;
  unsigned int Size() const { return this->NbPoints; }
  void Roll(const float minPoint[3], const float maxPoint[3]);
  void Add(const PointCloud& pointcloud, bool fixed = false, double currentTime = -1., bool roll = true)
  {
    this->Add(pointcloud.data(), pointcloud.size(), fixed, currentTime, roll);
  }
  void Add(const lsa_point_t* points, std::size_t count, bool fixed = false, double currentTime = -1., bool roll = true);

  // The reference builds a nanoflann kd-tree here; the MI355X path only materialises the
  // sub-map cloud, the device search grid is built by lsa_set_target.
  void BuildSubMap();
  void BuildSubMap(const float minPoint[3], const float maxPoint[3], int minNbPoints = -1);
  // Would BuildSubMap(minPoint, maxPoint, minNbPoints) return the sub-map that is there?  The sub-map is a
  // function of the map, of the range of outer voxels the box touches and of minNbPoints; this compares the
  // last two with what the current sub-map was built for (the caller knows the map has not changed since).
  bool SubMapBuiltFor(const float minPoint[3], const float maxPoint[3], int minNbPoints) const;
  // IsSubMapKdTreeValid(): false after every map modification until the next BuildSubMap
  bool IsSubMapValid() const { return this->SubMapValid && this->SubMapCount > 0; }
  const lsa_point_t* SubMapData() const { return this->SubMapPtr; }
  std::size_t SubMapSize() const { return this->SubMapCount; }
  // Where BuildSubMap writes: by default a vector of its own; with a provider (called with the number of points
  // it has to hold, returns the buffer) straight into the caller's memory -- the pinned staging buffer of the
  // device target, so that the sub-map is extracted once and never copied on the host.
  void SetSubMapStorage(std::function<lsa_point_t*(std::size_t)> provider) { this->SubMapStorage = std::move(provider); }
  void ClearOldPoints(double currentTime);

private:
  int GridSize = 50;
  double VoxelResolution = 10.;
  double LeafSize = 0.2;
  RollingVG Voxels;
  float VoxelGridPosition[3] = {0.f, 0.f, 0.f};
  unsigned int NbPoints = 0;
  unsigned int AddSerial = 0;
  PointCloud SubMapOwn;
  std::function<lsa_point_t*(std::size_t)> SubMapStorage;
  lsa_point_t* SubMapPtr = nullptr;
  std::size_t SubMapCount = 0;
  void BeginSubMap(std::size_t capacity);
  bool SubMapValid = false;
  bool SubMapBoxed = false;  // built by the bounding-box overload
  int SubMapLo[3] = {0, 0, 0}, SubMapHi[3] = {0, 0, 0}, SubMapMinNbPoints = 0;
  void VoxelRange(const float minPoint[3], const float maxPoint[3], int lo[3], int hi[3]) const;
  unsigned int MinFramesPerVoxel = 0;
  SamplingMode Sampling = SamplingMode::MAX_INTENSITY;
  double DecayingThreshold = -1;
  bool Ordered = true;
  // the voxels pred(outer index, voxel) keeps, in key order, appended to out
  template <typename Pred> void AppendOrdered(Pred pred, lsa_point_t* out, std::size_t& count) const;

  std::unique_ptr<Crew> AddCrew;
  std::vector<int> AddOut, AddIn;  // outer / leaf voxel index per point of the cloud being added (-1: outside)
  void AddParallel(const lsa_point_t* points, std::size_t count, bool fixed, double currentTime);
  struct AddTally
  {
    unsigned int inserted = 0;
    bool updated = false;
  };
  // the per-point body of Add() once the voxel indices are known
  inline void AddOne(const lsa_point_t& point, SamplingVG& outer, int idxIn, const int vi[3], const float centerIn[3], bool fixed, double currentTime,
                     unsigned int serial, AddTally& tally);
  int To1d(const int v[3]) const { return v[2] * GridSize * GridSize + v[1] * GridSize + v[0]; }
  void To3d(int id, int v[3]) const;
  void GridOrigin(float o[3]) const;
};

}  // namespace host
}  // namespace lsa
