// dliom_imu_window_add_pose must leave the window exactly as it was when the Schur complement of the marginalisation
// fails: same size, same newest state, and the same IMU samples usable for a retry without being counted
// twice.  marginalize_oldest() cannot be made to fail from outside (its inputs passed the solve a moment earlier), so
// this program compiles imu_window.cc by itself with DLIOM_TEST_HOOKS -- a seam the shipped library does not have.
#define DLIOM_TEST_HOOKS 1
#include "../../d-liom_amd/csrc/imu_window.cc"

#include <cstdio>

static int fail(const char* what) {
  std::printf("FAIL: %s\n", what);
  return 1;
}

int main() {
  dliom_imu_window_options o;
  if (dliom_imu_window_default_options(&o) != DLIOM_OK) return fail("default options");
  o.window_size = 3;
  dliom_imu_window* w = nullptr;
  if (dliom_imu_window_create(&o, &w) != DLIOM_OK) return fail("create");
  const double pose0[7] = {0, 0, 0, 1, 0, 0, 0}, v0[3] = {1, 0, 0}, b0[6] = {0, 0, 0, 0, 0, 0};
  if (dliom_imu_window_initialize(w, pose0, v0, b0) != DLIOM_OK) return fail("initialize");
  const double acc[3] = {0, 0, o.gravity}, gyr[3] = {0, 0, 0};
  double pose[7], vel[3], bias[6];
  int status = DLIOM_OK;
  int k = 1;
  // fill the window: no marginalisation yet
  for (; k < o.window_size; ++k) {
    for (int i = 0; i < 20; ++i)
      if (dliom_imu_window_add_imu(w, acc, gyr, 0.005) != DLIOM_OK) return fail("add_imu");
    const double matched[7] = {0.1 * k, 0, 0, 1, 0, 0, 0};
    status = dliom_imu_window_add_pose(w, matched, 0, pose, vel, bias);
    if (status != DLIOM_OK) return fail("add_pose while filling");
  }
  if (dliom_imu_window_size(w) != o.window_size) return fail("window not full");
  for (int i = 0; i < 20; ++i)
    if (dliom_imu_window_add_imu(w, acc, gyr, 0.005) != DLIOM_OK) return fail("add_imu");
  double before_pose[7], before_vel[3], before_bias[6];
  if (dliom_imu_window_state(w, 0, before_pose, before_vel, before_bias) != DLIOM_OK) return fail("state");
  const double matched[7] = {0.1 * k, 0, 0, 1, 0, 0, 0};
  dliom_test_fail_marginalize = 1;
  status = dliom_imu_window_add_pose(w, matched, 0, pose, vel, bias);
  if (status != DLIOM_ERR_SOLVER) return fail("failed marginalisation must return DLIOM_ERR_SOLVER");
  if (dliom_imu_window_size(w) != o.window_size) return fail("window size changed by a failed add_pose");
  double after_pose[7], after_vel[3], after_bias[6];
  if (dliom_imu_window_state(w, 0, after_pose, after_vel, after_bias) != DLIOM_OK) return fail("state after");
  if (std::memcmp(before_pose, after_pose, sizeof(before_pose)) != 0 || std::memcmp(before_vel, after_vel, sizeof(before_vel)) != 0 ||
      std::memcmp(before_bias, after_bias, sizeof(before_bias)) != 0)
    return fail("newest state changed by a failed add_pose");
  // the retry sees the same preintegration once: its result equals a window that never failed
  dliom_test_fail_marginalize = 0;
  status = dliom_imu_window_add_pose(w, matched, 0, pose, vel, bias);
  if (status != DLIOM_OK) return fail("retry");
  if (dliom_imu_window_size(w) != o.window_size) return fail("window size after the retry");
  dliom_imu_window* ref = nullptr;
  if (dliom_imu_window_create(&o, &ref) != DLIOM_OK || dliom_imu_window_initialize(ref, pose0, v0, b0) != DLIOM_OK) return fail("ref");
  double rpose[7], rvel[3], rbias[6];
  for (int kk = 1; kk <= k; ++kk) {
    for (int i = 0; i < 20; ++i) dliom_imu_window_add_imu(ref, acc, gyr, 0.005);
    const double m[7] = {0.1 * kk, 0, 0, 1, 0, 0, 0};
    if (dliom_imu_window_add_pose(ref, m, 0, rpose, rvel, rbias) != DLIOM_OK) return fail("ref add_pose");
  }
  if (std::memcmp(pose, rpose, sizeof(pose)) != 0 || std::memcmp(vel, rvel, sizeof(vel)) != 0 || std::memcmp(bias, rbias, sizeof(bias)) != 0)
    return fail("retry differs from a window that never failed");
  dliom_imu_window_destroy(w);
  dliom_imu_window_destroy(ref);
  std::printf("OK\n");
  return 0;
}
