"""Synthetic world, LiDAR scans and trajectory for tests and bench.py (numpy only).

World and scan model are the ones SURVEY.md §8(d) fixes (after the generator in the reference's
local_trajectory_builder_3d_test.cc:117-250): a closed scene so that every ray returns -- a 30 m
axis-aligned cube centred at the origin plus 100 spheres of radius 0.5 m placed at
10 * normalize(U(-1,1)^3), seed 42 -- scanned by a B-beam x A-azimuth spinning LiDAR with
elevations -15..+15 degrees, direction Rz(phi) * Ry(theta) * x, per-point relative time
t = -T (1 - k/(AB-1)) (last point 0), T = 0.1 s, and the corkscrew trajectory
p = (sin 4t, 1 - cos 4t, t), rotation 0.3 t about (1,-1,2)/sqrt(6).
"""
import numpy as np

CUBE_HALF = 15.0
NUM_BUBBLES = 100
BUBBLE_RADIUS = 0.5


def bubbles(seed=42):
    rng = np.random.RandomState(seed)
    v = rng.uniform(-1.0, 1.0, size=(NUM_BUBBLES, 3))
    return 10.0 * v / np.linalg.norm(v, axis=1, keepdims=True)


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    s = np.sin(0.5 * angle)
    return np.array([np.cos(0.5 * angle), axis[0] * s, axis[1] * s, axis[2] * s])


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def quat_to_matrix(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose_compose(a, b):
    """[t,q] (x) [t,q] in double (plain formulas; test scaffolding, not a parity path)."""
    Ra = quat_to_matrix(a[3:])
    t = Ra @ b[:3] + a[:3]
    q = quat_mul(a[3:], b[3:])
    q = q / np.linalg.norm(q)
    return np.concatenate([t, q])


def pose_inverse(a):
    q = np.array([a[3], -a[4], -a[5], -a[6]])
    t = -(quat_to_matrix(q) @ a[:3])
    return np.concatenate([t, q])


# p(t) = (R sin wt, R (1 - cos wt), t).  The reference test's corkscrew is R = 1 m, w = 4 rad/s (4 m/s on a 1 m
# circle: 16 m/s^2); set_trajectory(10.0, 0.4) is a vehicle-like arc (4 m/s, 1.6 m/s^2) for the streaming tool.
TRAJ = {"radius": 1.0, "omega": 4.0}


def set_trajectory(radius=1.0, omega=4.0):
    TRAJ["radius"], TRAJ["omega"] = float(radius), float(omega)


def trajectory_pose(t):
    """Corkscrew pose at time t (seconds)."""
    R, w = TRAJ["radius"], TRAJ["omega"]
    p = np.array([R * np.sin(w * t), R * (1.0 - np.cos(w * t)), t])
    q = quat_from_axis_angle([1.0, -1.0, 2.0], 0.3 * t)
    return np.concatenate([p, q])


def beam_directions(num_beams, num_azimuths):
    """Unit directions in the sensor frame, azimuth-major (a * B + b), plus relative times."""
    b = np.arange(num_beams)
    theta = np.deg2rad(-15.0 + 30.0 * b / max(num_beams - 1, 1))
    a = np.arange(num_azimuths)
    phi = 2.0 * np.pi * a / num_azimuths
    # Rz(phi) * Ry(theta) * x_hat ; Ry(theta) x = (cos th, 0, -sin th): positive theta looks down
    ct, st = np.cos(theta), np.sin(theta)
    cp, sp = np.cos(phi), np.sin(phi)
    d = np.stack([np.outer(cp, ct), np.outer(sp, ct), np.tile(-st, (num_azimuths, 1))], axis=-1)
    d = d.reshape(-1, 3)
    k = np.arange(num_beams * num_azimuths)
    rel_t = -0.1 * (1.0 - k / max(num_beams * num_azimuths - 1, 1))
    return d, rel_t


def cast(origin, dirs, centers=None):
    """Range along each ray from `origin` (inside the cube) to the nearest surface."""
    if centers is None:
        centers = bubbles()
    o = np.asarray(origin, dtype=np.float64)
    d = np.asarray(dirs, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_pos = (CUBE_HALF - o) / d
        t_neg = (-CUBE_HALF - o) / d
    t_box = np.where(d > 0, t_pos, np.where(d < 0, t_neg, np.inf)).min(axis=1)
    best = t_box
    for c in centers:
        oc = o - c
        bq = d @ oc
        cq = oc @ oc - BUBBLE_RADIUS ** 2
        disc = bq * bq - cq
        hit = disc > 0
        t = np.where(hit, -bq - np.sqrt(np.where(hit, disc, 0.0)), np.inf)
        t = np.where(t > 1e-9, t, np.inf)
        best = np.minimum(best, t)
    return best


def scan(pose, num_beams=64, num_azimuths=1024, noise_sigma=0.0, noise_seed=7, centers=None):
    """One scan taken at `pose` ([t,q] world<-sensor).  Returns float32 points in the SENSOR frame
    (n,3) and the per-point relative times."""
    dirs_s, rel_t = beam_directions(num_beams, num_azimuths)
    R = quat_to_matrix(pose[3:])
    dirs_w = dirs_s @ R.T
    rng_ = cast(pose[:3], dirs_w, centers)
    if noise_sigma > 0:
        rng_ = rng_ + np.random.RandomState(noise_seed).normal(0.0, noise_sigma, size=rng_.shape)
    pts = (dirs_s * rng_[:, None]).astype(np.float32)
    return pts, rel_t.astype(np.float32)


def transform_points(pose, pts):
    """float64 transform of points by [t,q] (scaffolding for building submaps)."""
    R = quat_to_matrix(pose[3:])
    return (np.asarray(pts, dtype=np.float64) @ R.T + pose[:3]).astype(np.float32)


def perturb_pose(pose, max_translation, max_angle_deg, seed=13):
    rng = np.random.RandomState(seed)
    dt = rng.uniform(-max_translation, max_translation, size=3)
    axis = rng.uniform(-1, 1, size=3)
    ang = np.deg2rad(rng.uniform(-max_angle_deg, max_angle_deg))
    return pose_compose(pose, np.concatenate([dt, quat_from_axis_angle(axis, ang)]))


def range_filter(pts, max_range):
    r = np.linalg.norm(pts.astype(np.float64), axis=1)
    return pts[r <= max_range]


# ---------------------------------------------------------------- motion-distorted scans + IMU (config 3)
AXIS = np.array([1.0, -1.0, 2.0]) / np.sqrt(6.0)
GRAVITY = np.array([0.0, 0.0, 9.80511])  # trajectory_builder_3d.lua:92; the IMU measures R^T (a + G)


def trajectory_velocity(t):
    R, w = TRAJ["radius"], TRAJ["omega"]
    return np.array([R * w * np.cos(w * t), R * w * np.sin(w * t), 1.0])


def trajectory_state(t):
    """[P(3), Q(4), V(3), Ba(3), Bg(3)] of the corkscrew at time t."""
    pose = trajectory_pose(t)
    return np.concatenate([pose, trajectory_velocity(t), np.zeros(6)])


def imu_samples(t0, t1, rate=200.0, noise=None, seed=11):
    """Specific force R^T (a + G) and body rate at t0, t0 + 1/rate, ..., t1 (inclusive)."""
    n = int(round((t1 - t0) * rate)) + 1
    ts = t0 + np.arange(n) / rate
    R, w = TRAJ["radius"], TRAJ["omega"]
    acc_w = np.stack([-R * w * w * np.sin(w * ts), R * w * w * np.cos(w * ts), np.zeros(n)], axis=1) + GRAVITY
    acc = np.stack([quat_to_matrix(quat_from_axis_angle(AXIS, 0.3 * t)).T @ a for t, a in zip(ts, acc_w)])
    gyr = np.tile(0.3 * AXIS, (n, 1))  # rotation about a fixed axis: body rate == world rate
    if noise is not None:
        rng = np.random.RandomState(seed)
        acc = acc + noise[0] * rng.normal(size=acc.shape)
        gyr = gyr + noise[1] * rng.normal(size=gyr.shape)
    return 1.0 / rate, acc, gyr


def cast_many(origins, dirs, centers=None):
    """cast() with one origin per ray."""
    if centers is None:
        centers = bubbles()
    o = np.asarray(origins, dtype=np.float64)
    d = np.asarray(dirs, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_pos = (CUBE_HALF - o) / d
        t_neg = (-CUBE_HALF - o) / d
    best = np.where(d > 0, t_pos, np.where(d < 0, t_neg, np.inf)).min(axis=1)
    for c in centers:
        oc = o - c
        bq = (d * oc).sum(axis=1)
        cq = (oc * oc).sum(axis=1) - BUBBLE_RADIUS ** 2
        disc = bq * bq - cq
        hit = disc > 0
        t = np.where(hit, -bq - np.sqrt(np.where(hit, disc, 0.0)), np.inf)
        t = np.where(t > 1e-9, t, np.inf)
        best = np.minimum(best, t)
    return best


def moving_scan(t_end, num_beams=64, num_azimuths=1024, centers=None):
    """A scan swept while the sensor flies the corkscrew: point k is measured at t_end + rel_t[k] in the
    sensor frame OF THAT INSTANT.  Returns float32 [x, y, z, rel_t] rows."""
    dirs_s, rel_t = beam_directions(num_beams, num_azimuths)
    ts = t_end + rel_t
    R, w = TRAJ["radius"], TRAJ["omega"]
    pos = np.stack([R * np.sin(w * ts), R * (1.0 - np.cos(w * ts)), ts], axis=1)
    ang = 0.3 * ts
    K = np.array([[0, -AXIS[2], AXIS[1]], [AXIS[2], 0, -AXIS[0]], [-AXIS[1], AXIS[0], 0]])
    # Rodrigues: R d = d + sin(a) K d + (1 - cos a) K K d
    Kd = dirs_s @ K.T
    KKd = Kd @ K.T
    dirs_w = dirs_s + np.sin(ang)[:, None] * Kd + (1.0 - np.cos(ang))[:, None] * KKd
    rng_ = cast_many(pos, dirs_w, centers)
    return np.concatenate([dirs_s * rng_[:, None], rel_t[:, None]], axis=1).astype(np.float32)
