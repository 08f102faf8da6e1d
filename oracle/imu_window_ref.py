"""TEST INFRASTRUCTURE -- not part of the product.  Independent double-precision restatement of the factors of
LocalTrajectoryBuilder3D::WindowOptimize (local_trajectory_builder_3d.cc:693-863) used to check
d-liom_amd/csrc/imu_window.cc: manifold IMU preintegration (Forster et al., the definition behind
gtsam::PreintegratedImuMeasurements), ImuFactor, bias BetweenFactor (:808-812), PriorFactor<Pose3> on the matched
pose with the reference's (t,t,t,r,r,r) sigma order (:94-101), initial priors (:712-745).  Solved as ONE batch
Gauss-Newton problem over every state (no window, no marginalisation), numerical Jacobians.
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


class Preintegration:
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


def retract(s, d):
    R, p, v, ba, bg = s
    return (R @ exp_so3(d[0:3]), p + d[3:6], v + d[6:9], ba + d[9:12], bg + d[12:15])


class BatchSmoother:
    """All states, all factors, Gauss-Newton to convergence after every new pose."""

    def __init__(self, opts):
        self.o = opts
        self.x, self.pre, self.pose_priors = [], [], []
        self.cur = None

    def initialize(self, pose7, vel, bias6):
        s = (quat_to_matrix(pose7[3:]), np.array(pose7[:3], float), np.array(vel, float), np.array(bias6[:3], float),
             np.array(bias6[3:], float))
        self.x, self.pre, self.pose_priors = [s], [], []
        self.prior0 = s
        self.cur = Preintegration(s[3], s[4], self.o)

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
            dR, dp, dv = P.corrected(a[3], a[4])
            raw = np.concatenate([log_so3(dR.T @ a[0].T @ b[0]),
                                  a[0].T @ (b[1] - a[1] - P.dt * a[2] - 0.5 * P.dt ** 2 * g) - dp,
                                  a[0].T @ (b[2] - a[2] - P.dt * g) - dv])
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
            step = np.linalg.solve(J.T @ J + 1e-12 * np.eye(n), -J.T @ r0)
            self.x = [retract(s, step[15 * i:15 * i + 15]) for i, s in enumerate(self.x)]
        s = self.x[-1]
        self.cur = Preintegration(s[3], s[4], self.o)
        return s
