// The reference's own known-answer tests, written against the C++ adapters
// (d-liom_amd/cpp/dliom_cartographer.h) the way the originals are written against cartographer:
//   real_time_correlative_scan_matcher_3d_test.cc:36-117, ceres_scan_matcher_3d_test.cc:34-100,
//   range_data_inserter_3d_test.cc:68-86.
// Built and run by tests/test_gpu_parity.py::test_cpp_adapters (needs a GPU).
#include <cmath>
#include <cstdio>

#include "../../d-liom_amd/cpp/dliom_cartographer.h"

using namespace dliom;
using mapping::HybridGrid;
using sensor::Vector3f;
using transform::Rigid3d;

static int g_failures = 0;
#define EXPECT(cond)                                                  \
  do {                                                                \
    if (!(cond)) {                                                    \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      ++g_failures;                                                   \
    }                                                                 \
  } while (0)

static const sensor::PointCloud kCloud = {{-3.f, 2.f, 0.f}, {-4.f, 2.f, 0.f}, {-5.f, 2.f, 0.f},
                                          {-6.f, 2.f, 0.f}, {-6.f, 3.f, 1.f}, {-6.f, 4.f, 2.f},
                                          {-7.f, 3.f, 1.f}};

static void FillGrid(HybridGrid* grid) {  // expected pose = translation (-1, 0, 0)
  for (const Vector3f& p : kCloud)
    grid->SetProbability(grid->GetCellIndex(Vector3f{p.x - 1.f, p.y, p.z}), 1.f);
}

static bool IsNearly(const Rigid3d& a, double tx, double ty, double tz, double eps) {
  const double dq = 2. * std::sqrt(a.rotation().x() * a.rotation().x() + a.rotation().y() * a.rotation().y() +
                                   a.rotation().z() * a.rotation().z());
  const double dt = std::sqrt(std::pow(a.translation().x() - tx, 2) + std::pow(a.translation().y() - ty, 2) +
                              std::pow(a.translation().z() - tz, 2));
  return std::sqrt(2. * dq * dq + dt * dt) <= eps * std::sqrt(5.);  // isApprox on the 4x4 affine
}

int main() {
  Context context(0);
  {  // RealTimeCorrelativeScanMatcher3DTest
    HybridGrid grid(&context, 0.1f);
    FillGrid(&grid);
    const mapping::scan_matching::RealTimeCorrelativeScanMatcher3D matcher(
        &context, {0.3, M_PI / 180., 1e-1, 1.});
    const double s = std::sin(0.4 / 180. * M_PI), c = std::cos(0.4 / 180. * M_PI);
    const Rigid3d inits[] = {
        Rigid3d::Translation({{-1., 0., 0.}}), Rigid3d::Translation({{-0.8, 0., 0.}}),
        Rigid3d::Translation({{-1., 0., -0.2}}), Rigid3d::Translation({{-0.9, -0.2, 0.2}}),
        Rigid3d({{-1., 0., 0.}}, {{c, s, 0., 0.}}), Rigid3d({{-1., 0., 0.}}, {{c, 0., s, 0.}}),
        Rigid3d({{-1., 0., 0.}}, {{c, 0., s, s}})};
    for (const Rigid3d& init : inits) {
      Rigid3d pose;
      const float score = matcher.Match(init, kCloud, grid, &pose);
      EXPECT(score > 0.f);
      EXPECT(IsNearly(pose, -1., 0., 0., 1e-3));
    }
  }
  {  // CeresScanMatcher3DTest
    HybridGrid grid(&context, 1.f);
    FillGrid(&grid);
    mapping::scan_matching::CeresScanMatcherOptions3D o;
    o.occupied_space_weight = {1.};
    o.translation_weight = 0.01;
    o.rotation_weight = 0.1;
    o.use_nonmonotonic_steps = true;
    o.max_num_iterations = 10;
    const mapping::scan_matching::CeresScanMatcher3D matcher(&context, o);
    const double starts[][3] = {{-1., 0., 0.}, {-0.8, 0., 0.}, {-1., 0., -0.2}, {-0.9, -0.2, 0.2}};
    for (const auto& t : starts) {
      const Rigid3d init = Rigid3d::Translation({{t[0], t[1], t[2]}});
      Rigid3d pose;
      mapping::scan_matching::Summary summary;
      matcher.Match(init.translation(), init, {{&kCloud, &grid}}, &pose, &summary);
      EXPECT(std::fabs(summary.final_cost) < 1e-2);
      EXPECT(IsNearly(pose, -1., 0., 0., 3e-2));
    }
  }
  {  // RangeDataInserter3DTest.InsertPointCloud
    HybridGrid grid(&context, 1.f);
    const mapping::RangeDataInserter3D inserter(&context, {0.7, 0.4, 1000});
    inserter.Insert(sensor::RangeData{{0.f, 0.f, -4.f}, {{-3.f, -1.f, 4.f}, {-2.f, 0.f, 4.f}, {-1.f, 1.f, 4.f},
                                                          {0.f, 2.f, 4.f}}, {}},
                    &grid);
    const std::vector<uint16_t> v = grid.values({{{0, 0, -4}}, {{0, 0, -3}}, {{-3, -1, 4}}, {{0, 2, 4}}, {{4, 4, 4}}});
    const uint16_t miss = dliom_probability_to_value(0.4f), hit = dliom_probability_to_value(0.7f);
    EXPECT(std::abs(int(v[0]) - int(miss)) <= 4 && std::abs(int(v[1]) - int(miss)) <= 4);
    EXPECT(std::abs(int(v[2]) - int(hit)) <= 4 && std::abs(int(v[3]) - int(hit)) <= 4);
    EXPECT(v[4] == 0);
  }
  {  // FastCorrelativeScanMatcher3DTest.CorrectPoseForMatch (fast_correlative_scan_matcher_3d_test.cc:134-163)
    const sensor::PointCloud cloud = {{4.f, 0.f, 0.f}, {4.5f, 0.f, 0.f}, {5.f, 0.f, 0.f}, {5.5f, 0.f, 0.f},
                                      {0.f, 4.f, 0.f}, {0.f, 4.5f, 0.f}, {0.f, 5.f, 0.f}, {0.f, 5.5f, 0.f},
                                      {0.f, 0.f, 4.f}, {0.f, 0.f, 4.5f}, {0.f, 0.f, 5.f}, {0.f, 0.f, 5.5f}};
    const float poses[][4] = {{0.3f, -0.5f, 0.2f, 0.f}, {-0.6f, 0.1f, -0.4f, 0.f}, {0.f, 0.65f, 0.35f, 0.f}};
    for (const auto& e : poses) {
      HybridGrid grid(&context, 0.05f);
      const mapping::RangeDataInserter3D inserter(&context, {0.7, 0.4, 5});
      const float c = std::cos(0.5f * e[3]), s = std::sin(0.5f * e[3]);
      sensor::PointCloud moved;
      for (const auto& p : cloud)  // Rz(theta) p + t
        moved.push_back({(1.f - 2.f * s * s) * p.x - 2.f * c * s * p.y + e[0], 2.f * c * s * p.x + (1.f - 2.f * s * s) * p.y + e[1],
                         p.z + e[2]});
      inserter.Insert(sensor::RangeData{{e[0], e[1], e[2]}, moved, {}}, &grid);
      const mapping::scan_matching::FastCorrelativeScanMatcher3D matcher(
          &context, grid, &grid, {{std::vector<float>(10, 0.f), e[3]}}, {6, 6, 0.1, 0.15, 0.8, 0.8, 0.3});
      mapping::scan_matching::TrajectoryNodeData data;
      data.high_resolution_point_cloud = cloud;
      data.low_resolution_point_cloud = cloud;
      data.rotational_scan_matcher_histogram.assign(10, 0.f);
      mapping::scan_matching::FastCorrelativeScanMatcher3D::Result result;
      EXPECT(matcher.Match(Rigid3d(), Rigid3d(), data, 0.1f, &result));
      EXPECT(result.score > 0.1f && result.rotational_score > 0.09f && result.low_resolution_score > 0.14f);
      EXPECT(IsNearly(result.pose_estimate, e[0], e[1], e[2], 0.05));
      data.low_resolution_point_cloud = {{42.f, 42.f, 42.f}};
      EXPECT(!matcher.Match(Rigid3d(), Rigid3d(), data, 0.1f, &result));
    }
  }
  std::printf(g_failures == 0 ? "ALL ADAPTER TESTS PASSED\n" : "%d FAILURES\n", g_failures);
  return g_failures == 0 ? 0 : 1;
}
