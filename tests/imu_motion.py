"""TEST INFRASTRUCTURE.  A random smooth rigid motion with closed-form derivatives, for the IMU-window tests
(tests/test_imu_fuzz.py): position = a sum of sinusoids per axis, attitude = Exp(phi(t)) with phi a sum of sinusoids,
hence velocity, acceleration and the BODY angular rate omega = J_r(phi) phi' in closed form, and the IMU of the
reference's convention (dliom.synth.imu_samples): specific force R^T (a + G), G = (0, 0, g)."""
import numpy as np
from scipy.spatial.transform import Rotation as Rot


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def _right_jacobian(w):
    t = np.linalg.norm(w)
    K = _skew(w)
    if t < 1e-6:
        return np.eye(3) - 0.5 * K + K @ K / 6.0
    return np.eye(3) - (1 - np.cos(t)) / t ** 2 * K + (t - np.sin(t)) / t ** 3 * (K @ K)


class RandomMotion:
    def __init__(self, seed, gravity=9.80511, translation_amplitude=1.0, rotation_amplitude=0.4, max_frequency=2.0, terms=3):
        rng = np.random.RandomState(seed)
        self.G = np.array([0.0, 0.0, gravity])
        self.pa = rng.uniform(-1, 1, (terms, 3)) * translation_amplitude
        self.pw = rng.uniform(0.3, max_frequency, (terms, 3)) * 2 * np.pi
        self.pp = rng.uniform(0, 2 * np.pi, (terms, 3))
        self.ra = rng.uniform(-1, 1, (terms, 3)) * rotation_amplitude
        self.rw = rng.uniform(0.3, max_frequency, (terms, 3)) * 2 * np.pi
        self.rp = rng.uniform(0, 2 * np.pi, (terms, 3))
        self.v0 = rng.uniform(-1, 1, 3)  # a constant drift on top

    def position(self, t):
        return (self.pa * np.sin(self.pw * t + self.pp)).sum(axis=0) + self.v0 * t

    def velocity(self, t):
        return (self.pa * self.pw * np.cos(self.pw * t + self.pp)).sum(axis=0) + self.v0

    def acceleration(self, t):
        return -(self.pa * self.pw ** 2 * np.sin(self.pw * t + self.pp)).sum(axis=0)

    def phi(self, t):
        return (self.ra * np.sin(self.rw * t + self.rp)).sum(axis=0)

    def phi_dot(self, t):
        return (self.ra * self.rw * np.cos(self.rw * t + self.rp)).sum(axis=0)

    def rotation(self, t):
        return Rot.from_rotvec(self.phi(t)).as_matrix()

    def body_rate(self, t):
        return _right_jacobian(self.phi(t)) @ self.phi_dot(t)

    def specific_force(self, t):
        return self.rotation(t).T @ (self.acceleration(t) + self.G)

    def pose7(self, t):
        q = Rot.from_rotvec(self.phi(t)).as_quat()  # x, y, z, w
        return np.concatenate([self.position(t), [q[3], q[0], q[1], q[2]]])

    def imu(self, t0, t1, rate, midpoint=True):
        """(dt, acc[n], gyr[n]) for the n = round((t1 - t0) rate) intervals of [t0, t1], sampled at their midpoints (or starts)."""
        n = int(round((t1 - t0) * rate))
        dt = (t1 - t0) / n
        ts = t0 + (np.arange(n) + (0.5 if midpoint else 0.0)) * dt
        return dt, np.stack([self.specific_force(t) for t in ts]), np.stack([self.body_rate(t) for t in ts])
