// LocalTrajectoryBuilder3D / ActiveSubmaps3D adapters (d-liom_amd/cpp/dliom_cartographer.h) driven like
// cartographer drives the reference's classes: AddImuData at 200 Hz, AddRangeData at 10 Hz
// (mapping/internal/3d/local_trajectory_builder_3d.h:83-111).  Input: a binary file written by
// tests/test_gpu_parity.py (scans with per-point times, IMU samples, the initial state); output: one line per
// MatchingResult with the local pose, compared there with the Python-driven chain.
//   g++ -std=c++17 ltb3d_adapter.cc -L../../d-liom_amd -ldliom
#include <cstdio>
#include <map>
#include <cstring>
#include <string>
#include <vector>

#include "../../d-liom_amd/cpp/dliom_cartographer.h"

using namespace dliom;

namespace {
// a recording metrics::FamilyFactory (cartographer/metrics/family_factory.h)
struct RecH : metrics::Histogram {
  std::vector<double> seen;
  void Observe(double v) override { seen.push_back(v); }
};
struct RecG : metrics::Gauge {
  int sets = 0;
  void Increment() override {}
  void Increment(double) override {}
  void Decrement() override {}
  void Decrement(double) override {}
  void Set(double) override { ++sets; }
};
template <typename Base, typename Impl>
struct RecFamily : metrics::Family<Base> {
  std::map<std::string, std::unique_ptr<Impl>> by_label;
  Base* Add(const std::map<std::string, std::string>& labels) override {
    std::string key;
    for (const auto& kv : labels) key += kv.first + "=" + kv.second + ";";
    by_label[key].reset(new Impl);
    return by_label[key].get();
  }
};
struct Recorder : metrics::FamilyFactory {
  std::map<std::string, std::unique_ptr<RecFamily<metrics::Histogram, RecH>>> histograms;
  std::map<std::string, std::unique_ptr<RecFamily<metrics::Gauge, RecG>>> gauges;
  std::map<std::string, size_t> buckets;
  metrics::Family<metrics::Gauge>* NewGaugeFamily(const std::string& name, const std::string&) override {
    gauges[name].reset(new RecFamily<metrics::Gauge, RecG>);
    return gauges[name].get();
  }
  metrics::Family<metrics::Histogram>* NewHistogramFamily(const std::string& name, const std::string&,
                                                          const metrics::Histogram::BucketBoundaries& b) override {
    histograms[name].reset(new RecFamily<metrics::Histogram, RecH>);
    buckets[name] = b.size();
    return histograms[name].get();
  }
};
}  // namespace

static bool read_all(FILE* f, void* p, size_t bytes) { return std::fread(p, 1, bytes, f) == bytes; }

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (f == nullptr) return 2;
  int32_t header[4];  // num_scans, points_per_scan, imu_per_scan, num_accumulated_range_data
  double init[16];    // pose7, velocity3, bias6
  if (!read_all(f, header, sizeof header) || !read_all(f, init, sizeof init)) return 2;
  Context context(0);
  mapping::LocalTrajectoryBuilderOptions3D options;
  std::memset(&options.front_end, 0, sizeof options.front_end);
  if (!read_all(f, &options.front_end, sizeof options.front_end)) return 2;
  Check(dliom_imu_window_default_options(&options.imu), "dliom_imu_window_default_options");
  double imu_noise[4];
  if (!read_all(f, imu_noise, sizeof imu_noise)) return 2;
  options.imu.acc_noise = imu_noise[0];
  options.imu.gyr_noise = imu_noise[1];
  options.imu.acc_bias_noise = imu_noise[2];
  options.imu.gyr_bias_noise = imu_noise[3];
  const bool gravity = argc > 2 && std::string(argv[2]) == "gravity";
  if (gravity) {  // dlio/config/basic_config_3d.lua:80: EstimateGravity + Pose3GravityFactor inside WindowOptimize
    options.imu.enable_gravity_factor = 1;
    options.imu.frames_for_online_gravity_estimate = 3;
    options.imu.window_size = 8;
  }
  options.num_accumulated_range_data = header[3];
  options.min_range = 1.f;
  options.max_range = 100.f;
  options.voxel_filter_size = 0.15f;
  options.scan_period = 0.1;
  Recorder recorder;  // LocalTrajectoryBuilder3D::RegisterMetrics (local_trajectory_builder_3d.h:113)
  mapping::LocalTrajectoryBuilder3D::RegisterMetrics(&recorder);
  mapping::LocalTrajectoryBuilder3D builder(&context, options, {"lidar"});
  builder.SetInitialState(transform::Rigid3d::FromArray(init), transform::Vector3d{{init[7], init[8], init[9]}}, init + 10);
  int64_t t = 0;
  int results = 0;
  for (int s = 0; s < header[0]; ++s) {
    std::vector<double> imu(static_cast<size_t>(header[2]) * 7);  // dt, acc3, gyr3
    if (!read_all(f, imu.data(), imu.size() * 8)) return 2;
    for (int k = 0; k < header[2]; ++k) {
      t += static_cast<int64_t>(imu[7 * k] * 1e7 + 0.5);
      sensor::ImuData d;
      d.time = t;
      std::memcpy(d.linear_acceleration, &imu[7 * k + 1], 24);
      std::memcpy(d.angular_velocity, &imu[7 * k + 4], 24);
      builder.AddImuData(d);
    }
    sensor::TimedPointCloudData cloud;
    cloud.time = t;
    cloud.origin = sensor::Vector3f{0.f, 0.f, 0.f};
    cloud.ranges.resize(static_cast<size_t>(header[1]));
    if (!read_all(f, cloud.ranges.data(), cloud.ranges.size() * 16)) return 2;
    std::unique_ptr<mapping::LocalTrajectoryBuilder3D::MatchingResult> r = builder.AddRangeData("lidar", cloud);
    if (r != nullptr) {
      const std::array<double, 7> p = r->local_pose.ToArray();
      std::printf("RESULT %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %zu %d\n", s, p[0], p[1], p[2], p[3], p[4], p[5], p[6],
                  r->range_data_in_local.returns.size(), r->insertion_result != nullptr ? 1 : 0);
      if (r->insertion_result != nullptr) {  // TrajectoryNode::Data::rotational_scan_matcher_histogram
        double sum = 0.;
        for (float v : r->insertion_result->rotational_scan_matcher_histogram) sum += v;
        std::printf("HISTOGRAM %d %zu %.9g\n", s, r->insertion_result->rotational_scan_matcher_histogram.size(), sum);
        std::printf("FILTERED %d %zu %zu\n", s, r->insertion_result->high_resolution_point_cloud.size(),
                    r->insertion_result->low_resolution_point_cloud.size());
        if (r->insertion_result->high_resolution_point_cloud.empty() || r->insertion_result->low_resolution_point_cloud.empty()) return 5;
        if (r->insertion_result->rotational_scan_matcher_histogram.size() != 120u || !(sum > 0.)) return 4;
      }
      ++results;
    }
  }
  std::fclose(f);
  {
    transform::Vector3d g{{0, 0, 0}};
    int64_t factors = 0;
    const bool valid = builder.GravityEstimate(&g, &factors);
    std::printf("GRAVITY valid %d factors %lld g %.17g %.17g %.17g\n", valid ? 1 : 0, static_cast<long long>(factors), g.v[0], g.v[1], g.v[2]);
  }
  const auto submaps = builder.active_submaps().submaps();
  std::printf("SUBMAPS %zu matching_index %d results %d\n", submaps.size(), builder.active_submaps().matching_index(), results);
  if (!submaps.empty()) {  // Submap3D::ToProto round trip: header fields back out of the serialized proto::Submap
    const std::string bytes = submaps[0]->ToProtoBytes(true);
    double pose[7];
    int32_t num = -1;
    int fin = -1;
    int64_t ho = 0, hs = 0, lo = 0, ls = 0;
    const int st = dliom_submap3d_from_proto(reinterpret_cast<const uint8_t*>(bytes.data()), static_cast<int64_t>(bytes.size()), 1,
                                             pose, &num, &fin, &ho, &hs, &lo, &ls);
    std::printf("SUBMAP_PROTO status %d bytes %zu num_range_data %d (%d) finished %d (%d) grids %lld %lld\n", st, bytes.size(), num,
                submaps[0]->num_range_data(), fin, submaps[0]->finished() ? 1 : 0, static_cast<long long>(hs),
                static_cast<long long>(ls));
    if (st != 0 || num != submaps[0]->num_range_data() || fin != (submaps[0]->finished() ? 1 : 0) || hs <= 0 || ls <= 0) return 3;
  }
  {
    const auto n = [&](const char* family, const char* label) {
      return recorder.histograms.at(family)->by_label.at(label)->seen.size();
    };
    const size_t ceres = n("mapping_internal_3d_local_trajectory_builder_costs", "scan_matcher=ceres;");
    const size_t score = n("mapping_internal_3d_local_trajectory_builder_scores", "scan_matcher=real_time_correlative;");
    const size_t dist = n("mapping_internal_3d_local_trajectory_builder_residuals", "component=distance;");
    const size_t angle = n("mapping_internal_3d_local_trajectory_builder_residuals", "component=angle;");
    const int latency = recorder.gauges.at("mapping_internal_3d_local_trajectory_builder_latency")->by_label.at("")->sets;
    std::printf("METRICS ceres %zu score %zu distance %zu angle %zu latency %d buckets %zu %zu %zu\n", ceres, score, dist, angle, latency,
                recorder.buckets.at("mapping_internal_3d_local_trajectory_builder_scores"),
                recorder.buckets.at("mapping_internal_3d_local_trajectory_builder_costs"),
                recorder.buckets.at("mapping_internal_3d_local_trajectory_builder_residuals"));
    const size_t want_score = options.front_end.use_online_correlative_scan_matching ? static_cast<size_t>(results) : 0u;
    if (ceres != static_cast<size_t>(results) || dist != ceres || angle != ceres || score != want_score || latency != results) return 6;
  }
  std::printf("LTB3D ADAPTER DONE\n");
  return 0;
}
