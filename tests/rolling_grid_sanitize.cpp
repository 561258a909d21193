// Driver of tests/test_host_sanitizers.py: the multi-threaded RollingGrid (Add and BuildSubMap on a crew of helper
// threads) against the single-threaded one, to be run under ThreadSanitizer and AddressSanitizer on the CPU.
#include "lsa_rolling_grid.h"
#include <cstdio>
#include <cstring>
#include <random>
using namespace lsa::host;
int main()
{
  RollingGrid g, ref;
  g.SetGridSize(16); ref.SetGridSize(16);
  g.SetVoxelResolution(6.); ref.SetVoxelResolution(6.);
  g.SetLeafSize(0.5); ref.SetLeafSize(0.5);
  g.SetAddThreads(4);
  std::mt19937 rng(7);
  std::normal_distribution<float> n(0.f, 14.f);
  for (int step = 0; step < 12; ++step)
  {
    std::vector<lsa_point_t> pts(6000);
    for (auto& p : pts) { p = lsa_point_t{}; p.x = n(rng) + step * 3.f; p.y = n(rng); p.z = n(rng) / 6.f; p.intensity = (float)(rng() % 255); p.time = step * 0.1; }
    g.WakeAddThreads();
    g.Add(pts, false, step * 0.1);
    ref.Add(pts, false, step * 0.1);
    const float lo[3] = {-20.f + step * 3.f, -20.f, -5.f}, hi[3] = {20.f + step * 3.f, 20.f, 5.f};
    g.BuildSubMap(lo, hi, 100);
    ref.BuildSubMap(lo, hi, 100);
    if (g.SubMapSize() != ref.SubMapSize() || memcmp(g.SubMapData(), ref.SubMapData(), g.SubMapSize() * sizeof(lsa_point_t)) != 0) { std::printf("MISMATCH at %d\n", step); return 1; }
  }
  std::printf("ok %zu points in the last sub-map\n", g.SubMapSize());
  return 0;
}
