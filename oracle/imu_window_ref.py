"""TEST INFRASTRUCTURE -- not part of the product.  Independent double-precision restatement of the factors of
LocalTrajectoryBuilder3D::WindowOptimize (local_trajectory_builder_3d.cc:693-863) used to check
d-liom_amd/csrc/imu_window.cc: manifold IMU preintegration (Forster et al., the definition behind
gtsam::PreintegratedImuMeasurements), ImuFactor, bias BetweenFactor (:808-812), PriorFactor<Pose3> on the matched
pose with the reference's (t,t,t,r,r,r) sigma order (:94-101), initial priors (:712-745).  Solved as ONE batch
Gauss-Newton problem over every state (no window, no marginalisation), numerical Jacobians.
The gravity factor (gravity_factor/gravity_factor.cc:10-33 with GTSAM 4.0.2's Unit3::basis / Unit3::error /
Rot3::rotate Jacobians restated) and GravityEstimator (gravity_factor/gravity_estimator.cc, in the tree: restated line
by line with numpy, quirks kept) + EstimateGravity's deque handling (local_trajectory_builder_3d.cc:1106-1154).
PARITY UNPINNED: GTSAM 4.0.2 is not in /root/reference and the reference holds no test at this boundary."""
import numpy as np
from scipy.spatial.transform import Rotation as Rot


def exp_so3(w):
    return Rot.from_rotvec(np.asarray(w, dtype=float)).as_matrix()


def log_so3(R):
    return Rot.from_matrix(R).as_rotvec()


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def right_jacobian(w):
    t = np.linalg.norm(w)
    K = skew(w)
    if t < 1e-6:
        return np.eye(3) - 0.5 * K + K @ K / 6.0
    return np.eye(3) - (1 - np.cos(t)) / t ** 2 * K + (t - np.sin(t)) / t ** 3 * (K @ K)


def quat_to_matrix(q):
    w, x, y, z = np.asarray(q, dtype=float) / np.linalg.norm(q)
    return Rot.from_quat([x, y, z, w]).as_matrix()


def right_jacobian_inverse(w):
    return np.linalg.inv(right_jacobian(w))


class TangentPreintegration:
    """gtsam::TangentPreintegration + PreintegratedImuMeasurements (GTSAM 4.0.2, what the reference links: README.MD:13-15
    sets no GTSAM_TANGENT_PREINTEGRATION flag and 4.0.x defaults to ON), restated independently of imu_window.cc: the
    vector [theta, p, v]; A = d(new)/d(old) by CENTRAL DIFFERENCES of the update itself (imu_window.cc has it in closed form),
    B and C in closed form; bias Jacobians H <- A H - [B | C]; covariance A S A^T + B Sa/dt B^T + C Sw/dt C^T + position block."""

    tangent = True

    def __init__(self, ba, bg, opts):
        self.ba, self.bg, self.o = np.array(ba, float), np.array(bg, float), opts
        self.dt = 0.0
        self.x = np.zeros(9)
        self.H_a, self.H_g = np.zeros((9, 3)), np.zeros((9, 3))
        self.cov = np.zeros((9, 9))

    @staticmethod
    def _update(x, a, w, h):
        th, p, v = x[0:3], x[3:6], x[6:9]
        R = exp_so3(th)
        wt = right_jacobian_inverse(th) @ w
        an = R @ a
        return np.concatenate([th + wt * h, p + v * h + 0.5 * h * h * an, v + an * h])

    def add(self, acc, gyr, h):
        a, w = np.asarray(acc, float) - self.ba, np.asarray(gyr, float) - self.bg
        A = np.zeros((9, 9))
        for c in range(9):
            d = np.zeros(9)
            d[c] = 1e-6
            A[:, c] = (self._update(self.x + d, a, w, h) - self._update(self.x - d, a, w, h)) / 2e-6
        R = exp_so3(self.x[0:3])
        B = np.zeros((9, 3))
        B[3:6] = 0.5 * h * h * R
        B[6:9] = h * R
        Cg = np.zeros((9, 3))
        Cg[0:3] = h * right_jacobian_inverse(self.x[0:3])
        self.cov = A @ self.cov @ A.T + B @ B.T * (self.o["acc_noise"] ** 2 / h) + Cg @ Cg.T * (self.o["gyr_noise"] ** 2 / h)
        self.cov[3:6, 3:6] += np.eye(3) * self.o["integration_sigma"] ** 2 * h
        self.H_a = A @ self.H_a - B
        self.H_g = A @ self.H_g - Cg
        self.x = self._update(self.x, a, w, h)
        self.dt += h

    @property
    def dR(self):
        return exp_so3(self.x[0:3])

    @property
    def dp(self):
        return self.x[3:6]

    @property
    def dv(self):
        return self.x[6:9]

    def corrected(self, ba, bg):
        x = self.x + self.H_a @ (ba - self.ba) + self.H_g @ (bg - self.bg)
        return exp_so3(x[0:3]), x[3:6], x[6:9]


def make_preintegration(ba, bg, opts):
    return TangentPreintegration(ba, bg, opts) if opts.get("tangent_preintegration", 0) else Preintegration(ba, bg, opts)


def imu_raw_residual(P, a, b, g):
    """The 9 unwhitened residuals of the IMU factor between states a and b (R, p, v, ba, bg)."""
    dR, dp, dv = P.corrected(a[3], a[4])
    if getattr(P, "tangent", False):  # NavState::localCoordinates of the predicted state at state b
        return np.concatenate([log_so3(b[0].T @ a[0] @ dR),
                               b[0].T @ (a[1] + P.dt * a[2] + 0.5 * P.dt ** 2 * g + a[0] @ dp - b[1]),
                               b[0].T @ (a[2] + P.dt * g + a[0] @ dv - b[2])])
    return np.concatenate([log_so3(dR.T @ a[0].T @ b[0]),
                           a[0].T @ (b[1] - a[1] - P.dt * a[2] - 0.5 * P.dt ** 2 * g) - dp,
                           a[0].T @ (b[2] - a[2] - P.dt * g) - dv])


class Preintegration:
    tangent = False

    def __init__(self, ba, bg, opts):
        self.ba, self.bg, self.o = np.array(ba, float), np.array(bg, float), opts
        self.dt = 0.0
        self.dR, self.dp, self.dv = np.eye(3), np.zeros(3), np.zeros(3)
        self.J_R_bg = np.zeros((3, 3))
        self.J_p_ba, self.J_p_bg = np.zeros((3, 3)), np.zeros((3, 3))
        self.J_v_ba, self.J_v_bg = np.zeros((3, 3)), np.zeros((3, 3))
        self.cov = np.zeros((9, 9))

    def add(self, acc, gyr, h):
        a, w = np.asarray(acc, float) - self.ba, np.asarray(gyr, float) - self.bg
        E, Jr, R = exp_so3(w * h), right_jacobian(w * h), self.dR
        A = np.eye(9)
        A[0:3, 0:3] = E.T
        A[3:6, 0:3] = -0.5 * h * h * R @ skew(a)
        A[3:6, 6:9] = h * np.eye(3)
        A[6:9, 0:3] = -h * R @ skew(a)
        B = np.zeros((9, 3))
        B[3:6] = 0.5 * h * h * R
        B[6:9] = h * R
        Cg = np.zeros((9, 3))
        Cg[0:3] = h * Jr
        self.cov = A @ self.cov @ A.T + B @ B.T * (self.o["acc_noise"] ** 2 / h) + Cg @ Cg.T * (self.o["gyr_noise"] ** 2 / h)
        self.cov[3:6, 3:6] += np.eye(3) * self.o["integration_sigma"] ** 2 * h
        RaxJ = R @ skew(a) @ self.J_R_bg
        self.J_p_ba = self.J_p_ba + self.J_v_ba * h - 0.5 * h * h * R
        self.J_p_bg = self.J_p_bg + self.J_v_bg * h - 0.5 * h * h * RaxJ
        self.J_v_ba = self.J_v_ba - h * R
        self.J_v_bg = self.J_v_bg - h * RaxJ
        self.J_R_bg = E.T @ self.J_R_bg - h * Jr
        self.dp = self.dp + self.dv * h + 0.5 * h * h * (R @ a)
        self.dv = self.dv + h * (R @ a)
        self.dR = R @ E
        self.dt += h

    def corrected(self, ba, bg):
        da, dg = ba - self.ba, bg - self.bg
        return self.dR @ exp_so3(self.J_R_bg @ dg), self.dp + self.J_p_ba @ da + self.J_p_bg @ dg, \
            self.dv + self.J_v_ba @ da + self.J_v_bg @ dg


def unit3_basis(n):
    """gtsam::Unit3::basis(): axis of the smallest |component| (x, then y, then z on ties)."""
    n = np.asarray(n, float)
    m = np.abs(n)
    axis = np.array([0.0, 0.0, 1.0])
    if m[0] <= m[1] and m[0] <= m[2]:
        axis = np.array([1.0, 0.0, 0.0])
    elif m[1] <= m[0] and m[1] <= m[2]:
        axis = np.array([0.0, 1.0, 0.0])
    b1 = np.cross(n, axis)
    b1 /= np.linalg.norm(b1)
    return np.stack([b1, np.cross(n, b1)], axis=1)  # 3 x 2


def gravity_factor(R, nZ, bRef, sigma):
    """Pose3GravityFactor::evaluateError: whitened error (2) and the factor's own 2 x 3 rotation Jacobian."""
    roll = np.arctan2(R[2, 1], R[2, 2])
    pitch = np.arctan2(-R[2, 0], np.hypot(R[2, 1], R[2, 2]))
    Rrp = Rot.from_euler("y", pitch).as_matrix() @ Rot.from_euler("x", roll).as_matrix()
    q = Rrp @ bRef
    Bp, Bq = unit3_basis(nZ), unit3_basis(q)
    e = Bp.T @ q + 1e-5
    H = (Bp.T @ Bq) @ (-Bq.T @ Rrp @ skew(bRef))
    return e / sigma, H / sigma


def tangent_basis(g0):
    a = g0 / np.linalg.norm(g0)
    tmp = np.array([0.0, 0.0, 1.0])
    if np.array_equal(a, tmp):
        tmp = np.array([1.0, 0.0, 0.0])
    b = tmp - a * (a @ tmp)
    b /= np.linalg.norm(b)
    return np.stack([b, np.cross(a, b)], axis=1)


def estimate_gravity_vector(frames, tlb, Vs, g_norm):
    """GravityEstimator::Estimate.  frames: list of (R, T, dt, dP, dV) -- a pose and the preintegration stored with it."""
    n = len(frames)
    if n < 3:
        return np.zeros(3), False
    A, b = np.zeros((3, 3)), np.zeros(3)
    for i in range(n - 1):
        Ri, Ti = frames[i][0], frames[i][1]
        Rj, Tj, dt, dP, dV = frames[i + 1]
        tA = np.vstack([Ri.T * dt * dt / 2, Ri.T * dt])
        tb = np.concatenate([dP + Ri.T @ Rj @ tlb - tlb - Ri.T @ (Tj - Ti) + dt * Vs[i], dV + Vs[i] - Ri.T @ Rj @ Vs[i + 1]])
        A += tA.T @ tA
        b += tA.T @ tb
    g = np.linalg.solve(A * 1000.0, b * 1000.0)
    if not abs(np.linalg.norm(g) - g_norm) < 0.5:
        return g, False
    g0 = g / np.linalg.norm(g) * g_norm
    A2, b2 = np.zeros((2, 2)), np.zeros(2)  # never cleared between the four rounds (gravity_estimator.cc:117-167)
    for _ in range(4):
        lxly = tangent_basis(g0)
        for i in range(n - 1):
            Ri, Ti = frames[i][0], frames[i][1]
            Rj, Tj, dt, dP, dV = frames[i + 1]
            tA = np.vstack([Ri.T * dt * dt / 2 @ lxly, Ri.T * dt @ lxly])
            tb = np.concatenate([dP + Ri.T @ Rj @ tlb - tlb - Ri.T * dt * dt / 2 @ g0 - Ri.T @ (Tj - Ti) + dt * Vs[i],
                                 dV - Ri.T * dt @ g0 + Vs[i] - Ri.T @ Rj @ Vs[i + 1]])
            A2 += tA.T @ tA
            b2 += tA.T @ tb
        A2 *= 1000.0
        b2 *= 1000.0
        dg = np.linalg.solve(A2, b2)
        g0 = g0 + lxly @ dg
        g0 = g0 / np.linalg.norm(g0) * g_norm
    return g0, bool(abs(np.linalg.norm(g0) - g_norm) < 0.2)


def retract(s, d):
    R, p, v, ba, bg = s
    return (R @ exp_so3(d[0:3]), p + d[3:6], v + d[6:9], ba + d[9:12], bg + d[12:15])


class BatchSmoother:
    """All states, all factors, Gauss-Newton to convergence after every new pose."""

    def __init__(self, opts):
        self.o = opts
        self.x, self.pre, self.pose_priors = [], [], []
        self.cur = None
        self.gravity = []            # (state index, nZ)
        self.g_frames, self.g_vs = [], []
        self.g_est, self.g_valid = np.zeros(3), False

    def _estimate_gravity(self, prev, running):
        """LocalTrajectoryBuilder3D::EstimateGravity (:1106-1154), including the in-place rotation of g_est_Vs_."""
        win = self.o["frames_for_online_gravity_estimate"]
        self.g_frames.append((prev[0], prev[1], running.dt, running.dp.copy(), running.dv.copy()))
        self.g_vs.append(prev[2].copy())
        if len(self.g_frames) <= win + 1:
            return False
        self.g_frames.pop(0)
        self.g_vs.pop(0)
        Rw, pw = self.g_frames[0][0], self.g_frames[0][1]
        tmp = []
        for i, (R, T, dt, dP, dV) in enumerate(self.g_frames):
            tmp.append((Rw.T @ R, Rw.T @ (T - pw), dt, dP, dV))
            self.g_vs[i] = R.T @ self.g_vs[i]
        g_B, ok = estimate_gravity_vector(tmp, np.asarray(self.o.get("lidar_in_imu_translation", (0, 0, 0)), float),
                                          self.g_vs, self.o["gravity"])
        if not ok:
            return False
        self.g_est = Rw @ (-g_B)
        return bool(self.g_est[2] + self.o["gravity"] < 0.5)

    def initialize(self, pose7, vel, bias6):
        s = (quat_to_matrix(pose7[3:]), np.array(pose7[:3], float), np.array(vel, float), np.array(bias6[:3], float),
             np.array(bias6[3:], float))
        self.x, self.pre, self.pose_priors = [s], [], []
        self.gravity, self.g_frames, self.g_vs, self.g_valid = [], [], [], False
        self.prior0 = s
        self.cur = make_preintegration(s[3], s[4], self.o)

    def add_imu(self, acc, gyr, dt):
        self.cur.add(acc, gyr, dt)

    def _predict(self, s, P):
        dR, dp, dv = P.corrected(s[3], s[4])
        g = np.array([0, 0, -self.o["gravity"]])
        return (s[0] @ dR, s[1] + P.dt * s[2] + 0.5 * P.dt ** 2 * g + s[0] @ dp, s[2] + P.dt * g + s[0] @ dv, s[3], s[4])

    def _residuals(self, x):
        o, r = self.o, []
        R0, p0, v0, ba0, bg0 = self.prior0
        r.append(log_so3(R0.T @ x[0][0]) / o["prior_pose_noise"])
        r.append((x[0][1] - p0) / o["prior_pose_noise"])
        r.append((x[0][2] - v0) / o["prior_velocity_sigma"])
        r.append((x[0][3] - ba0) / o["prior_bias_sigma"])
        r.append((x[0][4] - bg0) / o["prior_bias_sigma"])
        g = np.array([0, 0, -o["gravity"]])
        for i, P in enumerate(self.pre):
            a, b = x[i], x[i + 1]
            raw = imu_raw_residual(P, a, b, g)
            L = np.linalg.cholesky(P.cov + 1e-18 * np.eye(9))
            r.append(np.linalg.solve(L, raw))
            r.append((b[3] - a[3]) / (np.sqrt(P.dt) * o["acc_bias_noise"]))
            r.append((b[4] - a[4]) / (np.sqrt(P.dt) * o["gyr_bias_noise"]))
        for idx, Rm, pm, s_rot, s_trans in self.pose_priors:
            r.append(log_so3(Rm.T @ x[idx][0]) / s_rot)
            r.append(Rm.T @ (x[idx][1] - pm) / s_trans)
        return np.concatenate(r)

    def add_pose(self, matched7, is_drift=False, iterations=8):
        nxt = self._predict(self.x[-1], self.cur)
        if self.o.get("enable_gravity_factor", 0):
            self.g_valid = self._estimate_gravity(self.x[-1], self.cur)
            win = self.o["frames_for_online_gravity_estimate"]
            key = len(self.x)  # key_ of the new state
            if self.g_valid and key - win >= 0:
                self.gravity.append((key - win, self.g_est / np.linalg.norm(self.g_est)))
        self.x.append(nxt)
        self.pre.append(self.cur)
        o = self.o
        self.pose_priors.append((len(self.x) - 1, quat_to_matrix(matched7[3:]), np.array(matched7[:3], float),
                                 o["ceres_pose_noise_t_drift"] if is_drift else o["ceres_pose_noise_t"],
                                 o["ceres_pose_noise_r_drift"] if is_drift else o["ceres_pose_noise_r"]))
        n = 15 * len(self.x)
        for _ in range(iterations):
            r0 = self._residuals(self.x)
            J = np.zeros((len(r0), n))
            for c in range(n):
                d = np.zeros(15)
                d[c % 15] = 1e-6
                xp = list(self.x)
                xp[c // 15] = retract(self.x[c // 15], d)
                xm = list(self.x)
                xm[c // 15] = retract(self.x[c // 15], -d)
                J[:, c] = (self._residuals(xp) - self._residuals(xm)) / 2e-6
            # the gravity factors bring their own (the reference factor's) Jacobian
            bRef = np.array([0.0, 0.0, -1.0])
            rows_r, rows_J = [r0], [J]
            for idx, nZ in self.gravity:
                e, H = gravity_factor(self.x[idx][0], nZ, bRef, o["prior_gravity_noise"])
                Jg = np.zeros((2, n))
                Jg[:, 15 * idx:15 * idx + 3] = H
                rows_r.append(e)
                rows_J.append(Jg)
            r0, J = np.concatenate(rows_r), np.vstack(rows_J)
            step = np.linalg.solve(J.T @ J + 1e-12 * np.eye(n), -J.T @ r0)
            self.x = [retract(s, step[15 * i:15 * i + 15]) for i, s in enumerate(self.x)]
        s = self.x[-1]
        self.cur = make_preintegration(s[3], s[4], self.o)
        return s


class ReferenceRuleSmoother:
    """What LocalTrajectoryBuilder3D::WindowOptimize keeps between scans (:693-863): the graph grows until
    key_ == num_range_data, then it is reset to ONE state whose priors are the marginal covariances of X, V and B taken
    separately (:750-797; the cross-covariances between pose, velocity and bias are dropped there), with the gravity factor
    (EstimateGravity + Pose3GravityFactor, :772-782,819-831) when the options enable it.  Two idealisations of ISAM2:

      relinearize_threshold=None  a batch Gauss-Newton solve to convergence over every key since the last reset;
      relinearize_threshold=t     ISAM2's own rule as the reference parameterises it (:676-679, t = 0.1, relinearizeSkip 1):
                                  every key keeps a linearisation point theta and an increment delta, the estimate is
                                  theta (+) delta (calculateEstimate), `updates` (2: update(graph, values); update();)
                                  times per scan: move the KEYS (X(i), V(i), B(i) one by one) whose |delta|_inf exceeds t,
                                  linearise EVERY factor at the points, solve the linear problem exactly.  (What GTSAM adds on top -- the Bayes tree,
                                  the wildfire threshold on older keys' deltas, Cayley retraction -- is not restated.)

    Dense normal equations, block-sparse numeric Jacobians (every factor touches at most two states; the gravity factor
    brings its own Jacobian, as in the reference): independent of imu_window.cc's chain solver and closed forms."""

    def __init__(self, opts, num_range_data, relinearize_threshold=None, updates=2):
        self.o, self.n_reset = opts, int(num_range_data)
        self.thr, self.updates = relinearize_threshold, int(updates)

    def initialize(self, pose7, vel, bias6):
        s = (quat_to_matrix(pose7[3:]), np.array(pose7[:3], float), np.array(vel, float), np.array(bias6[:3], float),
             np.array(bias6[3:], float))
        o = self.o
        self.x, self.pre, self.pose_priors = [s], [], []
        self.delta = [np.zeros(15)]
        self.gravity, self.g_frames, self.g_vs, self.g_valid, self.g_est = [], [], [], False, np.zeros(3)
        self.gravity_factors = 0
        # prior on state 0: mean and the inverse square roots of the three blocks (pose tangent = (rotation, translation
        # in the body frame), like gtsam::Pose3::Logmap to first order)
        self.prior = (s, np.eye(6) / o["prior_pose_noise"], np.eye(3) / o["prior_velocity_sigma"], np.eye(6) / o["prior_bias_sigma"])
        self.cur = make_preintegration(s[3], s[4], o)
        self.resets = 0
        self.relinearizations = 0

    def add_imu(self, acc, gyr, dt):
        self.cur.add(acc, gyr, dt)

    def estimate(self, i=-1):
        return retract(self.x[i], self.delta[i])

    # ---- factors: (state indices, residual function of those states)
    def _factors(self):
        o = self.o
        g = np.array([0, 0, -o["gravity"]])
        s0, Wx, Wv, Wb = self.prior
        fs = [((0,), lambda a: np.concatenate([Wx @ np.concatenate([log_so3(s0[0].T @ a[0]), s0[0].T @ (a[1] - s0[1])]),
                                               Wv @ (a[2] - s0[2]), Wb @ np.concatenate([a[3] - s0[3], a[4] - s0[4]])]))]
        for i, P in enumerate(self.pre):
            L = np.linalg.cholesky(P.cov + 1e-18 * np.eye(9))

            def imu(a, b, P=P, L=L):
                raw = imu_raw_residual(P, a, b, g)
                return np.concatenate([np.linalg.solve(L, raw), (b[3] - a[3]) / (np.sqrt(P.dt) * o["acc_bias_noise"]),
                                       (b[4] - a[4]) / (np.sqrt(P.dt) * o["gyr_bias_noise"])])
            fs.append(((i, i + 1), imu))
        for idx, Rm, pm, s_rot, s_trans in self.pose_priors:
            fs.append(((idx,), lambda a, Rm=Rm, pm=pm, s_rot=s_rot, s_trans=s_trans:
                       np.concatenate([log_so3(Rm.T @ a[0]) / s_rot, Rm.T @ (a[1] - pm) / s_trans])))
        return fs

    def _normal_equations(self):
        n = 15 * len(self.x)
        H, b = np.zeros((n, n)), np.zeros(n)
        for idx, f in self._factors():
            st = [self.x[i] for i in idx]
            r0 = f(*st)
            J = np.zeros((len(r0), 15 * len(idx)))
            for k in range(len(idx)):
                for c in range(15):
                    d = np.zeros(15)
                    d[c] = 1e-6
                    sp, sm = list(st), list(st)
                    sp[k], sm[k] = retract(st[k], d), retract(st[k], -d)
                    J[:, 15 * k + c] = (f(*sp) - f(*sm)) / 2e-6
            cols = np.concatenate([np.arange(15 * i, 15 * i + 15) for i in idx])
            H[np.ix_(cols, cols)] += J.T @ J
            b[cols] += J.T @ r0
        bRef = np.array([0.0, 0.0, -1.0])
        for idx, nZ in self.gravity:  # the reference factor's own 2 x 3 Jacobian on the rotation block
            e, Hg = gravity_factor(self.x[idx][0], nZ, bRef, self.o["prior_gravity_noise"])
            H[15 * idx:15 * idx + 3, 15 * idx:15 * idx + 3] += Hg.T @ Hg
            b[15 * idx:15 * idx + 3] += Hg.T @ e
        return H, b

    def _solve(self, iterations):
        if self.thr is None:  # batch Gauss-Newton: the points ARE the estimates
            for _ in range(iterations):
                H, b = self._normal_equations()
                step = np.linalg.solve(H, -b)
                self.x = [retract(s, step[15 * i:15 * i + 15]) for i, s in enumerate(self.x)]
                if np.linalg.norm(step) < 1e-11:
                    break
            return
        for _ in range(iterations):
            for i in range(len(self.x)):
                part = np.zeros(15)
                for lo, hi in ((0, 6), (6, 9), (9, 15)):  # X(i), V(i), B(i): ISAM2 looks at every key by itself
                    if np.max(np.abs(self.delta[i][lo:hi])) > self.thr:
                        part[lo:hi] = self.delta[i][lo:hi]
                        self.delta[i][lo:hi] = 0.0
                        self.relinearizations += 1
                if np.any(part != 0.0):
                    self.x[i] = retract(self.x[i], part)
            H, b = self._normal_equations()
            d = np.linalg.solve(H, -b)
            self.delta = [d[15 * i:15 * i + 15].copy() for i in range(len(self.x))]

    def add_pose(self, matched7, is_drift=False, iterations=6):
        o = self.o
        gravity_on = bool(o.get("enable_gravity_factor", 0))
        updates = iterations if self.thr is None else self.updates
        prev = self.estimate()  # prev_state_ / prev_bias_: what the reference read after the previous scan
        if len(self.x) == self.n_reset:  # key_ == num_range_data (:750): reset, keep the three marginals apart
            H, _ = self._normal_equations()
            # marginal covariance of the newest key; the information spans 1e-8 (velocity prior) .. 1e10 (bias walk): invert
            # the symmetrically scaled matrix (unit diagonal) so that the dense inverse keeps its digits
            d = 1.0 / np.sqrt(np.diag(H))
            cov = (np.linalg.inv(H * d[:, None] * d[None, :]) * d[:, None] * d[None, :])[-15:, -15:]
            s = prev
            T = np.eye(6)
            T[3:6, 3:6] = s[0].T  # translation increment: world frame here, body frame in Pose3's tangent
            cov_x = T @ cov[0:6, 0:6] @ T.T
            def sqrt_inv(C):  # W with W^T W = C^-1, through the scaled matrix as above
                e = 1.0 / np.sqrt(np.diag(C))
                return np.linalg.cholesky(np.linalg.inv(C * e[:, None] * e[None, :]) * e[:, None] * e[None, :]).T
            self.prior = (s, sqrt_inv(cov_x), sqrt_inv(cov[6:9, 6:9]), sqrt_inv(cov[9:15, 9:15]))
            self.x, self.pre, self.pose_priors, self.delta, self.gravity = [s], [], [], [np.zeros(15)], []
            self.resets += 1
            if gravity_on:  # :772-782
                self.g_valid = BatchSmoother._estimate_gravity(self, prev, self.cur)
                if self.g_valid:
                    self.gravity.append((0, self.g_est / np.linalg.norm(self.g_est)))
                    self.gravity_factors += 1
                    self._solve(updates if self.thr is None else 1)  # "optimize once" (:788)
        nxt = BatchSmoother._predict(self, prev, self.cur)
        key = len(self.x)  # key_ of the new state
        self.x.append(nxt)
        self.delta.append(np.zeros(15))
        self.pre.append(self.cur)
        self.pose_priors.append((len(self.x) - 1, quat_to_matrix(matched7[3:]), np.array(matched7[:3], float),
                                 o["ceres_pose_noise_t_drift"] if is_drift else o["ceres_pose_noise_t"],
                                 o["ceres_pose_noise_r_drift"] if is_drift else o["ceres_pose_noise_r"]))
        if gravity_on:  # :819-831
            self.g_valid = BatchSmoother._estimate_gravity(self, prev, self.cur)
            win = o["frames_for_online_gravity_estimate"]
            if self.g_valid and key - win >= 0:
                self.gravity.append((key - win, self.g_est / np.linalg.norm(self.g_est)))
                self.gravity_factors += 1
        self._solve(updates)
        s = self.estimate()
        self.cur = make_preintegration(s[3], s[4], o)
        return s
