"""a16: the IMU-preintegration cost in the optimisation (d-liom_amd/csrc/imu_window.cc) against an independent
numpy restatement (oracle/imu_window_ref.py, batch Gauss-Newton over all states) and against the closed-form
corkscrew.  Host code: runs without a GPU.  PARITY UNPINNED (GTSAM absent from the reference tree)."""
import numpy as np
import pytest

OPT_NAMES = ("acc_noise", "gyr_noise", "acc_bias_noise", "gyr_bias_noise", "gravity", "integration_sigma",
             "prior_pose_noise", "prior_velocity_sigma", "prior_bias_sigma", "ceres_pose_noise_t", "ceres_pose_noise_r",
             "ceres_pose_noise_t_drift", "ceres_pose_noise_r_drift", "prior_gravity_noise")


@pytest.fixture(scope="module")
def dl():
    import dliom
    dliom.load_library()
    return dliom


def _feed(w, ref, k, T, synth, bias=(np.zeros(3), np.zeros(3)), noise=None):
    dt, acc, gyr = synth.imu_samples(T * (k - 1), T * k, 200.0, noise, seed=11 + k)
    for a, g in zip(acc[:-1], gyr[:-1]):
        w.add_imu(a + bias[0], g + bias[1], dt)
        if ref is not None:
            ref.add_imu(a + bias[0], g + bias[1], dt)


def _angle(qa, qb):
    return 2.0 * np.arccos(min(1.0, abs(float(np.dot(qa / np.linalg.norm(qa), qb / np.linalg.norm(qb))))))


def test_window_equals_batch_reference_while_nothing_is_marginalised(dl):
    from dliom import synth
    from oracle.imu_window_ref import BatchSmoother
    w = dl.ImuWindow(window_size=8, iterations=8)
    opts = {n: getattr(w.options, n) for n in OPT_NAMES}
    ref = BatchSmoother(opts)
    st = synth.trajectory_state(0.0)
    w.initialize(st[:7], st[7:10], np.zeros(6))
    ref.initialize(st[:7], st[7:10], np.zeros(6))
    rng = np.random.RandomState(4)
    T = 0.1
    for k in range(1, 6):
        _feed(w, ref, k, T, synth)
        matched = synth.perturb_pose(synth.trajectory_pose(T * k), 0.02, 0.1, seed=40 + k)  # a noisy scan match
        pose, vel, bias, status = w.add_pose(matched)
        R, p, v, ba, bg = ref.add_pose(matched)
        assert status == 0
        assert np.linalg.norm(pose[:3] - p) < 1e-7
        from scipy.spatial.transform import Rotation as Rot
        qr = Rot.from_matrix(R).as_quat()
        assert _angle(pose[3:], np.array([qr[3], qr[0], qr[1], qr[2]])) < 1e-7
        assert np.linalg.norm(vel - v) < 1e-6
        assert np.linalg.norm(bias - np.concatenate([ba, bg])) < 1e-7
    del rng


def test_fixed_lag_stays_close_to_the_batch_solution(dl):
    """12 scans through a 4-state window (8 states marginalised) vs the batch solution over all 13 states."""
    from dliom import synth
    from oracle.imu_window_ref import BatchSmoother
    w = dl.ImuWindow(window_size=4, iterations=2)
    opts = {n: getattr(w.options, n) for n in OPT_NAMES}
    ref = BatchSmoother(opts)
    st = synth.trajectory_state(0.0)
    w.initialize(st[:7], st[7:10], np.zeros(6))
    ref.initialize(st[:7], st[7:10], np.zeros(6))
    T = 0.1
    for k in range(1, 13):
        _feed(w, ref, k, T, synth)
        matched = synth.perturb_pose(synth.trajectory_pose(T * k), 0.02, 0.1, seed=70 + k)
        pose, vel, bias, status = w.add_pose(matched)
        R, p, v, ba, bg = ref.add_pose(matched, iterations=3)
        assert status == 0 and len(w) <= 4
        assert np.linalg.norm(pose[:3] - p) < 2e-3  # linearisation points of the marginalised factors differ
        assert np.linalg.norm(vel - v) < 2e-2


def test_smoothed_pose_beats_prediction_and_raw_match_on_the_corkscrew(dl):
    """Noisy IMU (the reference's imu block) + noisy matches (2 cm / 0.1 deg): the fused pose is closer to the truth
    than both the IMU prediction and the scan match it was given."""
    from dliom import synth
    w = dl.ImuWindow()
    st = synth.trajectory_state(0.0)
    w.initialize(st[:7], st[7:10], np.zeros(6))
    T = 0.1
    e_pred, e_match, e_fused = [], [], []
    for k in range(1, 31):
        _feed(w, None, k, T, synth, noise=(0.08, 0.004))
        pred, _ = w.predict()
        truth = synth.trajectory_pose(T * k)
        matched = synth.perturb_pose(truth, 0.02, 0.1, seed=200 + k)
        pose, vel, bias, status = w.add_pose(matched)
        assert status == 0
        if k > 5:
            e_pred.append(np.linalg.norm(pred[:3] - truth[:3]))
            e_match.append(np.linalg.norm(matched[:3] - truth[:3]))
            e_fused.append(np.linalg.norm(pose[:3] - truth[:3]))
    assert np.mean(e_fused) < np.mean(e_match) and np.mean(e_fused) < np.mean(e_pred), (np.mean(e_pred), np.mean(e_match), np.mean(e_fused))


def test_constant_gyro_bias_is_recovered(dl):
    from dliom import synth
    w = dl.ImuWindow()
    st = synth.trajectory_state(0.0)
    w.initialize(st[:7], st[7:10], np.zeros(6))
    true_bg = np.array([0.004, -0.003, 0.002])
    T = 0.1
    for k in range(1, 41):
        _feed(w, None, k, T, synth, bias=(np.zeros(3), true_bg))
        pose, vel, bias, status = w.add_pose(synth.trajectory_pose(T * k))
        assert status == 0
    assert np.linalg.norm(bias[3:] - true_bg) < 0.5 * np.linalg.norm(true_bg), bias


def test_failure_detection_and_argument_checks(dl):
    from dliom import synth
    w = dl.ImuWindow()
    st = synth.trajectory_state(0.0)
    with pytest.raises(Exception):
        w.add_imu([0, 0, 9.8], [0, 0, 0], 0.005)  # not initialised
    w.initialize(st[:7], st[7:10], np.zeros(6))
    with pytest.raises(Exception):
        w.add_pose(st[:7])  # no IMU since the last pose
    for _ in range(20):
        w.add_imu([0.0, 0.0, 9.80511], [0, 0, 0], 0.005)
    far = st[:7].copy()
    far[0] += 400.0  # a "match" 400 m away within 0.1 s: velocity beyond 30 m/s -> FailureDetection (:896-913)
    pose, vel, bias, status = w.add_pose(far, is_drift=False)
    assert status == dl.ERR_DIVERGED
    with pytest.raises(Exception):
        w.add_imu([0, 0, 9.8], [0, 0, 0], 0.005)  # ResetParams(): must be re-initialised
