// ORACLE (test infrastructure, NOT product code).
//
// CeresScanMatcher3D::Match restated on top of om_ceres.h:
//   .../scan_matching/ceres_scan_matcher_3d.cc:63-69   DENSE_QR forced
//   .../scan_matching/ceres_scan_matcher_3d.cc:71-123  problem assembly + Solve
//   optimization/ceres_pose.cc:23-44                   parameter blocks t[3], q[4]=(w,x,y,z)
#ifndef ORACLE_OM_CSM3D_H_
#define ORACLE_OM_CSM3D_H_

#include <cmath>
#include <cstdlib>
#include <memory>
#include <utility>
#include <vector>

#include "om_ceres.h"
#include "om_cost_functions.h"

namespace oracle {

struct CeresScanMatcherOptions3D {
  std::vector<double> occupied_space_weight;
  double translation_weight = 0;
  double rotation_weight = 0;
  bool only_optimize_yaw = false;
  bool use_nonmonotonic_steps = false;
  int max_num_iterations = 50;
  int num_threads = 1;
};

class CeresScanMatcher3D {
 public:
  using PointCloudAndHybridGridPointers =
      std::pair<const PointCloud*, const HybridGrid*>;

  explicit CeresScanMatcher3D(const CeresScanMatcherOptions3D& options)
      : options_(options) {}

  void Match(const Vec3d& target_translation, const Rigid3d& initial_pose_estimate,
             const std::vector<PointCloudAndHybridGridPointers>& clouds_and_grids,
             Rigid3d* pose_estimate, ceres_like::Summary* summary) const {
    ceres_like::Problem problem;
    problem.t[0] = initial_pose_estimate.translation.x;
    problem.t[1] = initial_pose_estimate.translation.y;
    problem.t[2] = initial_pose_estimate.translation.z;
    problem.q[0] = initial_pose_estimate.rotation.w;
    problem.q[1] = initial_pose_estimate.rotation.x;
    problem.q[2] = initial_pose_estimate.rotation.y;
    problem.q[3] = initial_pose_estimate.rotation.z;
    if (options_.only_optimize_yaw) {
      problem.rotation_parameterization.reset(
          new ceres_like::YawOnlyQuaternionParameterization);
    } else {
      problem.rotation_parameterization.reset(
          new ceres_like::QuaternionParameterization);
    }
    // CHECK_EQ(occupied_space_weight_size, clouds.size())
    if (options_.occupied_space_weight.size() != clouds_and_grids.size()) std::abort();
    for (size_t i = 0; i != clouds_and_grids.size(); ++i) {
      if (!(options_.occupied_space_weight[i] > 0.)) std::abort();  // CHECK_GT
      const PointCloud& cloud = *clouds_and_grids[i].first;
      const HybridGrid& grid = *clouds_and_grids[i].second;
      auto functor = std::make_shared<OccupiedSpaceCostFunction3D>(
          options_.occupied_space_weight[i] /
              std::sqrt(static_cast<double>(cloud.size())),
          cloud, grid);
      problem.blocks.push_back(ceres_like::MakeAutoDiffBlock34(
          functor, static_cast<int>(cloud.size())));
    }
    if (options_.translation_weight > 0.) {  // fork change, :104-110
      problem.blocks.push_back(ceres_like::MakeAutoDiffBlock3(
          std::make_shared<TranslationDeltaCostFunctor3D>(
              options_.translation_weight, target_translation)));
    }
    if (options_.rotation_weight > 0.) {  // :113-118
      problem.blocks.push_back(ceres_like::MakeAutoDiffBlock4(
          std::make_shared<RotationDeltaCostFunctor3D>(
              options_.rotation_weight, initial_pose_estimate.rotation)));
    }
    ceres_like::Options solver_options;
    solver_options.use_nonmonotonic_steps = options_.use_nonmonotonic_steps;
    solver_options.max_num_iterations = options_.max_num_iterations;
    ceres_like::Solve(solver_options, &problem, summary);
    *pose_estimate =
        Rigid3d(Vec3d(problem.t[0], problem.t[1], problem.t[2]),
                Quatd(problem.q[0], problem.q[1], problem.q[2], problem.q[3]));
  }

 private:
  const CeresScanMatcherOptions3D options_;
};

}  // namespace oracle

#endif  // ORACLE_OM_CSM3D_H_
