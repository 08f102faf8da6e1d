// ORACLE (test infrastructure, NOT product code).
//
// The three residual functors CeresScanMatcher3D stacks, restated as templates
// over the scalar type (double for cost-only evaluation, Jet<7> for Jacobians,
// exactly how ceres::AutoDiffCostFunction invokes them).
//   .../scan_matching/interpolated_grid.h:51-103   8-neighbour smoothstep blend (z, then y, then x)
//   .../scan_matching/interpolated_grid.h:106-146  lower-voxel selection in FLOAT, x2 = x1 + res (float add)
//   .../scan_matching/occupied_space_cost_function_3d.h:50-80  r_i = s * (1 - P(T p_i))
//   .../scan_matching/translation_delta_cost_functor_3d.h:39-45
//   .../scan_matching/rotation_delta_cost_functor_3d.h:43-66, common/math.h:74-81
#ifndef ORACLE_OM_COST_FUNCTIONS_H_
#define ORACLE_OM_COST_FUNCTIONS_H_

#include "om_hybrid_grid.h"
#include "om_jet.h"
#include "om_sensor.h"

namespace oracle {

class InterpolatedGrid {
 public:
  explicit InterpolatedGrid(const HybridGrid& grid) : grid_(grid) {}

  template <typename T>
  T GetProbability(const T& x, const T& y, const T& z) const {
    // interpolated_grid.h:123-139 -- cell of the point (double -> float cast),
    // then step down where the centre lies above the coordinate; comparison is
    // float-centre vs double-coordinate.
    const double xs = ScalarPart(x), ys = ScalarPart(y), zs = ScalarPart(z);
    Vec3f lower = grid_.GetCenterOfCell(grid_.GetCellIndex(Vec3f(
        static_cast<float>(xs), static_cast<float>(ys), static_cast<float>(zs))));
    if (lower.x > xs) lower.x -= grid_.resolution();
    if (lower.y > ys) lower.y -= grid_.resolution();
    if (lower.z > zs) lower.z -= grid_.resolution();
    // interpolated_grid.h:108-120
    const double x1 = lower.x, y1 = lower.y, z1 = lower.z;
    const double x2 = lower.x + grid_.resolution();
    const double y2 = lower.y + grid_.resolution();
    const double z2 = lower.z + grid_.resolution();

    // interpolated_grid.h:56-72
    const Vec3i i1 = grid_.GetCellIndex(Vec3f(
        static_cast<float>(x1), static_cast<float>(y1), static_cast<float>(z1)));
    const double q111 = grid_.GetProbability(i1);
    const double q112 = grid_.GetProbability(i1 + Vec3i(0, 0, 1));
    const double q121 = grid_.GetProbability(i1 + Vec3i(0, 1, 0));
    const double q122 = grid_.GetProbability(i1 + Vec3i(0, 1, 1));
    const double q211 = grid_.GetProbability(i1 + Vec3i(1, 0, 0));
    const double q212 = grid_.GetProbability(i1 + Vec3i(1, 0, 1));
    const double q221 = grid_.GetProbability(i1 + Vec3i(1, 1, 0));
    const double q222 = grid_.GetProbability(i1 + Vec3i(1, 1, 1));

    // interpolated_grid.h:74-102
    const T nx = (x - x1) / (x2 - x1);
    const T ny = (y - y1) / (y2 - y1);
    const T nz = (z - z1) / (z2 - z1);
    const T nxx = nx * nx;
    const T nxxx = nx * nxx;
    const T nyy = ny * ny;
    const T nyyy = ny * nyy;
    const T nzz = nz * nz;
    const T nzzz = nz * nzz;
    const T q11 = (q111 - q112) * nzzz * 2. + (q112 - q111) * nzz * 3. + q111;
    const T q12 = (q121 - q122) * nzzz * 2. + (q122 - q121) * nzz * 3. + q121;
    const T q21 = (q211 - q212) * nzzz * 2. + (q212 - q211) * nzz * 3. + q211;
    const T q22 = (q221 - q222) * nzzz * 2. + (q222 - q221) * nzz * 3. + q221;
    const T q1 = (q11 - q12) * nyyy * 2. + (q12 - q11) * nyy * 3. + q11;
    const T q2 = (q21 - q22) * nyyy * 2. + (q22 - q21) * nyy * 3. + q21;
    return (q1 - q2) * nxxx * 2. + (q2 - q1) * nxx * 3. + q1;
  }

 private:
  const HybridGrid& grid_;
};

// occupied_space_cost_function_3d.h:34-85
class OccupiedSpaceCostFunction3D {
 public:
  OccupiedSpaceCostFunction3D(double scaling_factor, const PointCloud& cloud,
                              const HybridGrid& grid)
      : scaling_factor_(scaling_factor), cloud_(cloud), interpolated_grid_(grid) {}

  int num_residuals() const { return static_cast<int>(cloud_.size()); }

  // translation[3], rotation[4] = (w,x,y,z); the quaternion is NOT normalised.
  template <typename T>
  bool operator()(const T* translation, const T* rotation, T* residual) const {
    const Rigid3<T> transform(
        Vec3<T>(translation[0], translation[1], translation[2]),
        Quat<T>(rotation[0], rotation[1], rotation[2], rotation[3]));
    for (size_t i = 0; i < cloud_.size(); ++i) {
      const Vec3<T> world = transform * cloud_[i].template cast<T>();
      const T probability =
          interpolated_grid_.GetProbability(world.x, world.y, world.z);
      residual[i] = scaling_factor_ * (1. - probability);
    }
    return true;
  }

 private:
  const double scaling_factor_;
  const PointCloud& cloud_;
  const InterpolatedGrid interpolated_grid_;
};

// translation_delta_cost_functor_3d.h
class TranslationDeltaCostFunctor3D {
 public:
  TranslationDeltaCostFunctor3D(double scaling_factor, const Vec3d& target)
      : scaling_factor_(scaling_factor), x_(target.x), y_(target.y), z_(target.z) {}
  template <typename T>
  bool operator()(const T* translation, T* residual) const {
    residual[0] = scaling_factor_ * (translation[0] - x_);
    residual[1] = scaling_factor_ * (translation[1] - y_);
    residual[2] = scaling_factor_ * (translation[2] - z_);
    return true;
  }

 private:
  const double scaling_factor_, x_, y_, z_;
};

// rotation_delta_cost_functor_3d.h
class RotationDeltaCostFunctor3D {
 public:
  RotationDeltaCostFunctor3D(double scaling_factor, const Quatd& target)
      : scaling_factor_(scaling_factor) {
    inv_[0] = target.w;
    inv_[1] = -target.x;
    inv_[2] = -target.y;
    inv_[3] = -target.z;
  }
  template <typename T>
  bool operator()(const T* q, T* residual) const {
    // common/math.h:74-81 with z = target^-1 (double), w = q.
    const double* z = inv_;
    T d[4];
    d[0] = z[0] * q[0] - z[1] * q[1] - z[2] * q[2] - z[3] * q[3];
    d[1] = z[0] * q[1] + z[1] * q[0] + z[2] * q[3] - z[3] * q[2];
    d[2] = z[0] * q[2] - z[1] * q[3] + z[2] * q[0] + z[3] * q[1];
    d[3] = z[0] * q[3] + z[1] * q[2] - z[2] * q[1] + z[3] * q[0];
    residual[0] = scaling_factor_ * d[1];
    residual[1] = scaling_factor_ * d[2];
    residual[2] = scaling_factor_ * d[3];
    return true;
  }

 private:
  const double scaling_factor_;
  double inv_[4];
};

}  // namespace oracle

#endif  // ORACLE_OM_COST_FUNCTIONS_H_
