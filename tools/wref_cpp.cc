// W-ref through the C++ adapters, timed in C++ (round 6): what a cartographer process gets from the drop-in -- the
// reference's LocalTrajectoryBuilder3D call sequence (AddImuData at 200 Hz, AddRangeData at 10 Hz:
// mapping/internal/3d/local_trajectory_builder_3d.h:83-111) on d-liom_amd/cpp/dliom_cartographer.h, no Python between
// the calls.  tools/wref_full.py times the same chain through the ctypes mirror, whose per-call overhead (~15 calls a scan)
// is part of its figure; this harness is the one without it.  Input: the binary stream tools/wref_cpp.py writes.
//   g++ -std=c++17 -O2 wref_cpp.cc -L../d-liom_amd -ldliom
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../d-liom_amd/cpp/dliom_cartographer.h"

using namespace dliom;

static bool read_all(FILE* f, void* p, size_t bytes) { return std::fread(p, 1, bytes, f) == bytes; }

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (f == nullptr) return 2;
  int32_t header[6];  // scans, imu samples per scan, warm-up scans, histogram size, reserved x 2
  double init[16];    // pose7, velocity3, bias6
  float scalars[4];   // voxel_filter_size, min_range, max_range, scan_period
  mapping::LocalTrajectoryBuilderOptions3D options;
  if (!read_all(f, header, sizeof header) || !read_all(f, init, sizeof init) || !read_all(f, &options.front_end, sizeof options.front_end) ||
      !read_all(f, &options.imu, sizeof options.imu) || !read_all(f, scalars, sizeof scalars))
    return 2;
  options.keep_imu_window_size = true;  // the stream's own window options, as written
  options.voxel_filter_size = scalars[0];
  options.min_range = scalars[1];
  options.max_range = scalars[2];
  options.scan_period = scalars[3];
  options.num_accumulated_range_data = 1;
  options.rotational_histogram_size = header[3];
  Context context(0);
  mapping::LocalTrajectoryBuilder3D builder(&context, options, {"lidar"});
  // (the stream begins in motion and its initial state is the state at its first instant: the graph starts there)
  builder.SetInitialState(transform::Rigid3d::FromArray(init), transform::Vector3d{{init[7], init[8], init[9]}}, init + 10, true);
  const int scans = header[0], per = header[1], warmup = header[2];
  std::vector<std::vector<double>> imus(static_cast<size_t>(scans));
  std::vector<sensor::TimedPointCloudData> clouds(static_cast<size_t>(scans));
  int64_t t = 0;
  std::vector<std::vector<int64_t>> imu_times(static_cast<size_t>(scans));
  for (int s = 0; s < scans; ++s) {  // everything is read before the clock starts
    int32_t n = 0;
    if (!read_all(f, &n, 4)) return 2;
    imus[s].resize(static_cast<size_t>(per) * 7);
    if (!read_all(f, imus[s].data(), imus[s].size() * 8)) return 2;
    for (int k = 0; k < per; ++k) {
      t += static_cast<int64_t>(imus[s][7 * k] * 1e7 + 0.5);
      imu_times[s].push_back(t);
    }
    clouds[s].time = t;
    clouds[s].origin = sensor::Vector3f{0.f, 0.f, 0.f};
    clouds[s].ranges.resize(static_cast<size_t>(n));
    if (!read_all(f, clouds[s].ranges.data(), clouds[s].ranges.size() * 16)) return 2;
  }
  std::fclose(f);
  // "pinned": what a caller gains who keeps its point clouds in page-locked memory (dliom_host_register) -- the scan's
  // upload inside dliom_add_range_data is then one asynchronous DMA instead of the runtime's staged copy of pageable memory
  const bool pinned = argc > 3 && std::strcmp(argv[3], "pinned") == 0;
  if (pinned)
    for (auto& c : clouds) Check(dliom_host_register(context.get(), c.ranges.data(), c.ranges.size() * 16), "dliom_host_register");
  std::vector<double> ms;
  int results = 0, inserted = 0;
  int64_t read_backs0 = 0, read_backs1 = 0;
  double last_pose[7] = {0};
  std::vector<double> poses;  // every result's pose, for the Python side's comparison
  for (int s = 0; s < scans; ++s) {
    if (s == warmup) Check(dliom_ctx_read_backs(context.get(), &read_backs0), "read_backs");
#ifdef DLIOM_ADAPTER_STAGE_TIMES
    if (s == warmup) {
      for (int i = 0; i < 16; ++i) stage_times::table()[i] = 0.0;
      stage_times::last() = std::chrono::steady_clock::now();
    }
#endif
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < per; ++k) {
      sensor::ImuData d;
      d.time = imu_times[s][k];
      std::memcpy(d.linear_acceleration, &imus[s][7 * k + 1], 24);
      std::memcpy(d.angular_velocity, &imus[s][7 * k + 4], 24);
      builder.AddImuData(d);
    }
    std::unique_ptr<mapping::LocalTrajectoryBuilder3D::MatchingResult> r = builder.AddRangeData("lidar", clouds[s]);
    const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (s >= warmup) ms.push_back(dt);
    if (r != nullptr) {
      ++results;
      if (r->insertion_result != nullptr) ++inserted;
      const std::array<double, 7> p = r->local_pose.ToArray();
      std::copy(p.begin(), p.end(), last_pose);
      poses.insert(poses.end(), p.begin(), p.end());
    } else {
      poses.insert(poses.end(), 7, 0.0);
    }
  }
  Check(dliom_ctx_read_backs(context.get(), &read_backs1), "read_backs");
  double sum = 0;
  for (double v : ms) sum += v;
  std::vector<double> sorted = ms;
  std::sort(sorted.begin(), sorted.end());
  std::printf("{\"harness\": \"C++ adapters (tools/wref_cpp.cc)\", \"scans_timed\": %zu, \"scans_per_s\": %.3f, \"mean_ms\": %.5f, "
              "\"p50_ms\": %.5f, \"p99_ms\": %.5f, \"max_ms\": %.5f, \"results\": %d, \"inserted\": %d, \"read_backs_per_scan\": %.3f, "
              "\"histogram_host_fallbacks\": %lld, \"last_pose\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g]}\n",
              ms.size(), 1e3 * ms.size() / sum, sum / ms.size(), sorted[sorted.size() / 2], sorted[(sorted.size() * 99) / 100],
              sorted.back(), results, inserted, static_cast<double>(read_backs1 - read_backs0) / ms.size(),
              static_cast<long long>(builder.histogram_host_fallbacks()), last_pose[0], last_pose[1], last_pose[2], last_pose[3],
              last_pose[4], last_pose[5], last_pose[6]);
#ifdef DLIOM_ADAPTER_STAGE_TIMES
  {
    static const char* const names[11] = {"caller + AddImuData", "dliom_add_range_data", "front_end_match_cloud", "window_optimize",
                                          "result + resize", "histogram begin", "download returns (transformed)", "front_end_insert",
                                          "download filtered clouds", "histogram finish", "assemble"};
    std::fprintf(stderr, "stages, us per scan over the %d timed scans:", scans - warmup);
    for (int i = 0; i < 11; ++i) std::fprintf(stderr, " %s %.1f;", names[i], stage_times::table()[i] / (scans - warmup));
    std::fprintf(stderr, "\n");
  }
#endif
  if (pinned)
    for (auto& c : clouds) Check(dliom_host_unregister(context.get(), c.ranges.data()), "dliom_host_unregister");
  if (argc > 2) {  // poses for the comparison with the Python-driven chain
    FILE* o = std::fopen(argv[2], "wb");
    if (o != nullptr) {
      std::fwrite(poses.data(), 8, poses.size(), o);
      std::fclose(o);
    }
  }
  return 0;
}
