"""The oracle's restatement of Ceres 1.13's Levenberg-Marquardt (oracle/src/om_ceres.h) against an INDEPENDENT
trust-region solver ("pinned only by the reference's 3e-2 KAT; same iteration count compares two
restatements by the same author").  scipy.optimize.least_squares minimises the SAME stacked residual vector --
occupied-space residuals of both grids from the oracle's (KAT-pinned) cost function, the translation and rotation delta
residuals written out here from translation_delta_cost_functor_3d.h:39-45 / rotation_delta_cost_functor_3d.h:43-54 --
over a 6-dof local parameterisation of its own, with scipy's numerical Jacobian, to machine convergence.  Ceres stops
early (function_tolerance 1e-6), so the oracle's result must (i) not undercut the converged minimum, (ii) lie within
1e-4 relative of it in cost (observed 1e-5) and (iii) within 1e-3 m / 1e-4 rad of the minimiser in pose (observed: 3e-4 m, 5e-7 rad --
the valley is flat where Ceres' function tolerance stops it) -- 30 x tighter than the KAT."""
import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation as Rot

from helpers import DEFAULT_CSM, build_oracle_submap


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def _stacked_residuals(orc, opts, target_t, init7, clouds_and_grids, t, q):
    rows = []
    for (pts, grid), w in zip(clouds_and_grids, opts["occupied_space_weight"]):
        scaling = w / np.sqrt(float(len(pts)))  # ceres_scan_matcher_3d.cc:94-101
        r, _, _ = orc.occupied_space_evaluate(grid, pts, scaling, t, q, jacobians=False)
        rows.append(r)
    rows.append(opts["translation_weight"] * (np.asarray(t) - np.asarray(target_t)))
    qi = np.array([init7[3], -init7[4], -init7[5], -init7[6]])  # init_q^-1
    rows.append(opts["rotation_weight"] * _qmul(qi, q)[1:])
    return np.concatenate(rows)


@pytest.mark.parametrize("seed", [3, 8])
def test_oracle_lm_lands_where_an_independent_solver_lands(orc, seed):
    from dliom import synth
    og_hi = build_oracle_submap(orc, 0.1, num_scans=5, beams=16, azimuths=256)
    og_lo = build_oracle_submap(orc, 0.45, num_scans=5, beams=16, azimuths=256)
    truth = synth.trajectory_pose(0.5)
    pts, _ = synth.scan(truth, 16, 128)
    hi, lo = pts[::8], pts[::5]  # ~250 and ~400 points: the sizes the reference's adaptive filters hand the matcher
    init = synth.perturb_pose(truth, 0.05, 0.3, seed=seed)
    target_t = init[:3]
    cg = [(hi, og_hi), (lo, og_lo)]
    got = orc.csm3d_match(DEFAULT_CSM, target_t, init, cg)

    def unpack(x):
        q = Rot.from_rotvec(x[3:]).as_quat()  # x, y, z, w
        return init[:3] + x[:3], _qmul(np.array([q[3], q[0], q[1], q[2]]), init[3:])

    def fun(x):
        t, q = unpack(x)
        return _stacked_residuals(orc, DEFAULT_CSM, target_t, init, cg, t, q)

    sol = least_squares(fun, np.zeros(6), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12, diff_step=1e-7, max_nfev=400)
    cost_min = 0.5 * float(np.sum(sol.fun ** 2))
    t_min, q_min = unpack(sol.x)
    # the oracle evaluates the same objective at its own result
    r_or = _stacked_residuals(orc, DEFAULT_CSM, target_t, init, cg, got["pose"][:3], got["pose"][3:])
    cost_or = 0.5 * float(np.sum(r_or ** 2))
    assert abs(cost_or - got["final_cost"]) <= 1e-9 * cost_or, (cost_or, got["final_cost"])  # same objective as Summary::final_cost
    assert cost_or >= cost_min * (1.0 - 1e-9), (cost_or, cost_min)          # (i) nothing undercuts the converged minimum
    assert cost_or <= cost_min * (1.0 + 1e-4), (cost_or, cost_min)          # (ii) Ceres' early stop leaves <= 1e-4 of it (observed 1e-5)
    dt = float(np.linalg.norm(got["pose"][:3] - t_min))
    dq = float(2.0 * np.arccos(min(1.0, abs(float(np.dot(got["pose"][3:] / np.linalg.norm(got["pose"][3:]), q_min / np.linalg.norm(q_min)))))))
    assert dt <= 1e-3 and dq <= 1e-4, (dt, dq)                               # (iii) and the pose is the minimiser's
    assert got["initial_cost"] > cost_or                                      # (it did move)
