"""
CPU oracle for the batched MPC rollout path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy/SciPy fp64 *restatement* of the reference's algorithm for the
hot path.  It is the checker, never the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``catenary-..._amd/`` imports it, and the product path
raises when the HIP library is missing instead of falling back to this file.

Pinning status
--------------
* reference-defined pieces (Rodrigues rotation, transform_catenary, solve_catenary,
  the tension rule, the symbolic equations, the scaler, the RK4/Simpson update, the
  Euler update, the 18-feature map, the trajectory generator) are pinned by the
  golden vectors in ``tests/golden/`` which were generated in the build container by
  importing ``/root/reference/main_fun.py`` / sympy-lambdifying the reference's CSV
  rows (``tools/make_golden.py``).
* ``pympc.models.catenary.Catenary`` is an un-vendored third-party dependency whose
  source is absent from the reference snapshot (empty ``pympc/`` directory, version
  unpinned, no reference test holds an output of it): the ``Catenary`` class below
  follows the catenary physics that *is* in the repo
  (``models/catenary-3d-visualization/src/catenary_model.py:10-20``,
  ``main_fun.py:418-431``, ``models/catenary_3d.py:13-14``).  **parity unpinned**
  for that one callable.
* the closed-loop rollout, cost and arg-min are build-defined (the reference has no
  MPC solver); they are defined HERE and the HIP kernel is held to this definition.

All ``file:line`` citations are relative to ``/root/reference``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# A2 -- symbolic equations (saved_models/equations_d{theta,gamma}_dt.csv, sympy_format)
# --------------------------------------------------------------------------------------

_EXPR_NS = {
    "sin": np.sin, "cos": np.cos, "tanh": np.tanh, "exp": np.exp, "log": np.log,
    "sqrt": np.sqrt, "Abs": np.abs, "abs": np.abs,
    "square": np.square, "neg": np.negative,
    # cluster_run/train_dynamics.py:28-30 custom operators
    "safe_log": lambda x: np.log(np.abs(x) + 1e-5),
    "safe_sqrt": lambda x: np.sqrt(np.abs(x)),
}


class SymbolicModel:
    """``.predict(X(n,F)) -> (n,)`` like a PySRRegressor restricted to one chosen row.

    The expression string is the ``sympy_format`` column of the reference's equation
    CSV; it is evaluated with Python's own expression evaluator over a NumPy
    namespace (independent of the product's bytecode compiler).
    """

    def __init__(self, sympy_format: str, n_features: int = 18, variable_names: Optional[Sequence[str]] = None):
        self.expr = sympy_format.strip()
        self.n_features = n_features
        self.variable_names = list(variable_names) if variable_names else None   # dd_cluster.py:160-168
        self._code = compile(self.expr, "<sympy_format>", "eval")

    def __call__(self, xs: Sequence) -> np.ndarray:
        """xs: sequence of per-feature arrays (already scaled), len >= n_features."""
        ns = dict(_EXPR_NS)
        for i in range(len(xs)):
            ns[f"x{i}"] = xs[i]
        if self.variable_names:
            for nm, x in zip(self.variable_names, xs):
                ns[nm] = x
        with np.errstate(all="ignore"):
            out = eval(self._code, {"__builtins__": {}}, ns)  # noqa: S307 (trusted fixture text)
        return out

    def predict(self, X: np.ndarray) -> np.ndarray:
        X = np.asarray(X, dtype=np.float64)
        cols = [X[:, i] for i in range(X.shape[1])]
        out = self(cols)
        return np.broadcast_to(np.asarray(out, dtype=np.float64), (X.shape[0],)).copy()


# --------------------------------------------------------------------------------------
# A4 -- 18-feature map (simply.py:15-41; main_fun.py:167-193 is the same minus _prev)
# --------------------------------------------------------------------------------------

def extract_features_gen1(P0, P1, V1, time, theta, gamma):
    """P0,P1 in metres (the reference divides mm by 1000 before this point), V1 raw."""
    P0 = np.asarray(P0, float); P1 = np.asarray(P1, float); V1 = np.asarray(V1, float)
    time = np.asarray(time, float)
    A1 = np.stack([np.gradient(V1[:, j], time) for j in range(3)], axis=1)   # :20-23
    rel_vec = P1 - P0                                                        # :25
    nr = np.linalg.norm(rel_vec, axis=1, keepdims=True)
    unit_rel = rel_vec / (nr + 1e-8)                                         # :26
    tension = np.clip(nr, 1e-5, 10)                                          # :27
    dot_product = np.sum(V1 * unit_rel, axis=1, keepdims=True)               # :29
    norm_v1 = np.linalg.norm(V1, axis=1, keepdims=True) + 1e-8               # :30
    angle_proj = np.clip(dot_product / norm_v1, -1, 1)                       # :31
    theta = np.asarray(theta, float).reshape(-1, 1)
    gamma = np.asarray(gamma, float).reshape(-1, 1)
    theta_prev = np.roll(theta, 1); gamma_prev = np.roll(gamma, 1)           # :35-36
    theta_prev[0] = theta[0]; gamma_prev[0] = gamma[0]                       # :37-38
    return np.hstack([P1, V1, A1, unit_rel, tension, angle_proj, theta, gamma,
                      theta_prev, gamma_prev])                               # :41


def scale_features(X, mean, scale):
    """sklearn StandardScaler.transform: (x - mean_) / scale_ (simply.py:57-58)."""
    return (np.asarray(X, float) - np.asarray(mean, float)) / np.asarray(scale, float)


# --------------------------------------------------------------------------------------
# A3 -- open-loop integrators
# --------------------------------------------------------------------------------------

def rk4_replay(predict: Callable[[np.ndarray], np.ndarray], x_input, time, y0):
    """simulate_rk4_theta_gamma.py:52-68, statement by statement."""
    x_input = np.asarray(x_input, float); time = np.asarray(time, float)
    y = [float(y0)]
    for i in range(1, len(time)):
        dt = time[i] - time[i - 1]
        x0 = x_input[i - 1]
        x1 = x_input[i]
        f = lambda x: predict(x.reshape(1, -1))[0]                           # :59
        k1 = f(x0)
        k2 = f((x0 + x1) / 2)
        k3 = f((x0 + x1) / 2)
        k4 = f(x1)
        y.append(y[-1] + (dt / 6) * (k1 + 2 * k2 + 2 * k3 + k4))             # :66
    return np.array(y)


def euler_replay(predict_theta, predict_gamma, X, time_array, theta_0, gamma_0):
    """main_fun.py:735-764 integrate_theta_gamma."""
    X = np.asarray(X, float); time_array = np.asarray(time_array, float)
    n = len(time_array)
    th = np.zeros(n); ga = np.zeros(n)
    th[0] = theta_0; ga[0] = gamma_0
    for i in range(1, n):
        dt = time_array[i] - time_array[i - 1]
        th[i] = th[i - 1] + predict_theta(X[i - 1:i])[0] * dt
        ga[i] = ga[i - 1] + predict_gamma(X[i - 1:i])[0] * dt
    return th, ga


def double_euler_replay(dd_theta, dd_gamma, time, theta_0, gamma_0):
    """test_cluster.py:110-129: angular velocities from zero by Euler, then angles by Euler."""
    dd = [np.asarray(dd_theta, float), np.asarray(dd_gamma, float)]
    time = np.asarray(time, float)
    out = []
    for d, y0 in zip(dd, (theta_0, gamma_0)):
        dot = np.zeros_like(d)
        for i in range(1, len(time)):
            dt = time[i] - time[i - 1]
            dot[i] = dot[i - 1] + d[i - 1] * dt
        est = np.zeros_like(d)
        est[0] = y0
        for i in range(1, len(time)):
            dt = time[i] - time[i - 1]
            est[i] = est[i - 1] + dot[i - 1] * dt
        out.append(est)
    return out[0], out[1]


def trapezoid_replay(dd_theta, dd_gamma, time, theta_0, gamma_0):
    """dd_cluster.py:221-230: cumulative_trapezoid(initial=0) then theta0 + cumsum(diff(time) * dot[1:])."""
    from scipy.integrate import cumulative_trapezoid
    time = np.asarray(time, float)
    out = []
    for d, y0 in zip((dd_theta, dd_gamma), (theta_0, gamma_0)):
        dot = cumulative_trapezoid(np.asarray(d, float), time, initial=0)
        est = y0 + np.cumsum(np.diff(time) * dot[1:])
        out.append(np.insert(est, 0, y0))
    return out[0], out[1]


# --------------------------------------------------------------------------------------
# A5 -- catenary parameter and tension
# --------------------------------------------------------------------------------------

C_LO_DEFAULT, C_HI_DEFAULT = 1e-6, 10.0          # main_fun.py:425 bracket


def _f_catenary(C, l, dH, L):
    return C ** 2 * (L ** 2 - dH ** 2) - 4 * (np.sinh(0.5 * l * C)) ** 2     # main_fun.py:423


def solve_catenary_scalar(l, delta_H, L, c_lo=C_LO_DEFAULT, c_hi=C_HI_DEFAULT):
    """main_fun.py:421-428: scipy brentq on [1e-6, 10], any exception -> nan."""
    from scipy.optimize import root_scalar
    try:
        with np.errstate(all="ignore"):
            sol = root_scalar(lambda C: _f_catenary(C, l, delta_H, L),
                              bracket=[c_lo, c_hi], method="brentq")
        return sol.root
    except Exception:
        return np.nan


def solve_catenary_ref(l, delta_H, L, c_lo=C_LO_DEFAULT, c_hi=C_HI_DEFAULT):
    """main_fun.py:418-431 (np.vectorize over the scalar brentq)."""
    return np.vectorize(lambda a, b, c: solve_catenary_scalar(a, b, c, c_lo, c_hi),
                        otypes=[float])(l, delta_H, L)


def solve_catenary_vec(l, delta_H, L, c_lo=C_LO_DEFAULT, c_hi=C_HI_DEFAULT):
    """Vectorised solver with brentq's *contract* (same root to ~1e-15 rel, same nan set).

    nan  <=>  f(c_lo) * f(c_hi) > 0  (scipy brentq raises ValueError -> except -> nan);
    f(c_lo)==0 -> c_lo; f(c_hi)==0 -> c_hi; otherwise the unique sign change in the
    bracket, found by safeguarded Newton on  h(u) = sinh(u) - r u,  u = l C / 2,
    r = sqrt(L^2 - dH^2) / l, started right of the root (h convex => monotone).
    Used for large K where 38 us/brentq is too slow; tests pin it to
    ``solve_catenary_ref`` at 1e-11 and on the nan set exactly.
    """
    l, dH, L = np.broadcast_arrays(np.asarray(l, float), np.asarray(delta_H, float),
                                   np.asarray(L, float))
    with np.errstate(all="ignore"):
        flo = _f_catenary(c_lo, l, dH, L)
        fhi = _f_catenary(c_hi, l, dH, L)
        ok = ~(flo * fhi > 0)
        ok &= np.isfinite(flo) & np.isfinite(fhi)
        r = np.sqrt(L ** 2 - dH ** 2) / l
        # upper bound of the root from sinh(u)/u >= 1 + u^2/6 + u^4/120
        u = np.sqrt(np.maximum(60.0 * (-1.0 / 6.0 + np.sqrt(1.0 / 36.0 + (r - 1.0) / 30.0)), 0.0))
        u = np.where(ok & np.isfinite(u) & (u > 0), u, 1.0)
        for _ in range(60):
            h = np.sinh(u) - r * u
            hp = np.cosh(u) - r
            un = u - h / hp
            un = np.where(np.isfinite(un) & (un > 0), un, u)
            done = np.abs(un - u) <= 4e-16 * np.abs(un)
            u = un
            if np.all(done | ~ok):
                break
        C = 2.0 * u / l
        C = np.where(flo == 0, c_lo, C)
        C = np.where(fhi == 0, c_hi, C)
        C = np.where(ok, C, np.nan)
    return C


def cable_tension(l, C, L, cable_wet_weight):
    """main_fun.py:302-305."""
    w_per_unit_length = cable_wet_weight / L
    with np.errstate(all="ignore"):
        T = (w_per_unit_length * l) / (2 * np.sinh(C * l / 2))
    return np.where(np.isnan(T), w_per_unit_length * l / 2, T)


# --------------------------------------------------------------------------------------
# A6 -- Rodrigues, augmented catenary
# --------------------------------------------------------------------------------------

def rodrigues_rotation(vector, axis, angle_rad):
    """main_fun.py:18-35."""
    axis = axis / np.linalg.norm(axis)
    return (vector * np.cos(angle_rad)
            + np.cross(axis, vector) * np.sin(angle_rad)
            + axis * np.dot(axis, vector) * (1 - np.cos(angle_rad)))


class Catenary:
    """Stand-in for ``pympc.models.catenary.Catenary`` (source absent -> parity unpinned).

    ``Catenary(length=3., reference_frame='ENU')`` (catenary.py:10);
    ``catenary(a, b) -> tuple`` whose ``[3]`` is an ``(M,3)`` array from a to b or
    ``None`` (catenary.py:25-29, main_fun.py:64-69).

    Shape law: ``(cosh(C x) - 1)/C`` in the vertical plane through a,b
    (catenary_model.py:10-12) with C from main_fun.py:418-431 (arc length L);
    no curve (``None``) when that solve fails (taut / outside bracket), cf.
    models/catenary_3d.py:13-14 and main_fun.py:67-69 where callers substitute the
    straight segment.  Returned tuple (build-defined): ``(C, sag, x_low, points)``.
    """

    def __init__(self, length=3.0, reference_frame="ENU", n_points=32,
                 c_lo=C_LO_DEFAULT, c_hi=C_HI_DEFAULT):
        if reference_frame not in ("ENU", "NED"):
            raise ValueError("reference_frame must be 'ENU' or 'NED'")
        self.length = float(length)
        self.reference_frame = reference_frame
        self.up = 1.0 if reference_frame == "ENU" else -1.0
        self.n_points = int(n_points)
        self.c_lo, self.c_hi = c_lo, c_hi

    def __call__(self, a, b):
        a = np.asarray(a, float); b = np.asarray(b, float)
        rel = b - a
        l = math.sqrt(rel[0] * rel[0] + rel[1] * rel[1])
        dH = self.up * rel[2]
        C = solve_catenary_scalar(l, dH, self.length, self.c_lo, self.c_hi)
        if not np.isfinite(C):
            return (None, None, None, None)
        M = self.n_points
        x0 = 0.5 * l - math.atanh(dH / self.length) / C
        ch0 = math.cosh(C * x0)
        pts = np.empty((M, 3))
        for j in range(M):
            t = j / (M - 1)
            up_j = (math.cosh(C * (l * t - x0)) - ch0) / C
            pts[j, 0] = a[0] + t * rel[0]
            pts[j, 1] = a[1] + t * rel[1]
            pts[j, 2] = a[2] + self.up * up_j
        sag = (ch0 - 1.0) / C          # drop of the lowest point below a (if 0<=x0<=l)
        return (C, sag, x0, pts)


def compute_catenary_3d(p0, p1, rope_length, num_points):
    """models/catenary_3d.py:5-39 (compute_catenary_3D), the catenary generator the reference itself holds: straight
    np.linspace when the rope is not longer than the distance (:13-14); otherwise the fixed point on the catenary
    parameter from a = half the distance, at most 100 rounds, stop at |a_new - a| < 1e-6 (:16-24); z lowered by
    a cosh(half / a) - a cosh(x / a) along the chord (:26-37).  Pinned: tests/golden/kat_catenary_3d.npz holds the
    outputs of the reference's own function."""
    p0 = np.asarray(p0, float); p1 = np.asarray(p1, float)
    d = p1 - p0
    direct = np.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)
    if rope_length <= direct:
        return np.linspace(p0, p1, num_points)
    half = direct / 2
    a = half
    for _ in range(100):
        a_new = a * rope_length / (2 * a * np.sinh(direct / (2 * a)))
        stop = abs(a_new - a) < 1e-6
        a = a_new
        if stop:
            break
    off = a * np.cosh(half / a)
    t = np.arange(num_points) / (num_points - 1)
    pts = p0[None, :] + d[None, :] * t[:, None]
    pts[:, 2] -= off - a * np.cosh((t * direct - half) / a)
    return pts


class Catenary3D:
    """catenary_fn built on compute_catenary_3d (what main_fun.py:63-69 expects: [3] = points)."""

    def __init__(self, length=3.0, num_points=100):
        self.length, self.num_points = float(length), int(num_points)

    def __call__(self, a, b):
        return (None, None, None, compute_catenary_3d(a, b, self.length, self.num_points))


def transform_catenary(point_A, point_B, catenary_fn, theta_rad, gamma_rad):
    """main_fun.py:38-111, statement by statement (returns the 4-tuple the code returns)."""
    point_A = np.asarray(point_A, float); point_B = np.asarray(point_B, float)

    def compute_catenary(start, end):
        output = catenary_fn(start, end)
        if output[3] is not None:
            return output[3]
        return np.array([start, end])

    original_catenary = compute_catenary(point_A, point_B)
    connection_vector = point_B - point_A
    xy_projection = connection_vector.copy()
    xy_projection[2] = 0
    if np.linalg.norm(xy_projection) < 1e-9:
        xy_projection = np.array([1., 0., 0.])
    else:
        xy_projection /= np.linalg.norm(xy_projection)
    z_axis = np.array([0, 0, 1])
    theta_axis = np.cross(xy_projection, z_axis)
    if np.linalg.norm(theta_axis) < 1e-9:
        theta_axis = np.array([0., 1., 0.])
    else:
        theta_axis /= np.linalg.norm(theta_axis)
    rotated_B = point_A + rodrigues_rotation(connection_vector, theta_axis, theta_rad)
    theta_rotated_catenary = compute_catenary(point_A, rotated_B)
    theta_aligned_catenary = np.array([
        point_A + rodrigues_rotation(pt - point_A, theta_axis, -theta_rad)
        for pt in theta_rotated_catenary])
    gamma_axis = point_B - point_A
    gamma_axis = gamma_axis / np.linalg.norm(gamma_axis)
    final_catenary = np.array([
        point_A + rodrigues_rotation(pt - point_A, gamma_axis, gamma_rad)
        for pt in theta_aligned_catenary])
    return original_catenary, theta_rotated_catenary, theta_aligned_catenary, final_catenary


def lowest_point(points, up=1.0):
    """fully_augmented_catenary.py:21-22: argmin z (ENU; NED flips the sign)."""
    idx = int(np.argmin(up * points[:, 2]))
    return idx, points[idx]


def theta_gamma_axes(rel):
    """The two rotation axes of main_fun.py:75-89,102-103 for connection vector rel."""
    xy = np.array([rel[0], rel[1], 0.0])
    nxy = np.linalg.norm(xy)
    xy = np.array([1., 0., 0.]) if nxy < 1e-9 else xy / nxy
    th = np.cross(xy, np.array([0., 0., 1.]))
    nth = np.linalg.norm(th)
    th = np.array([0., 1., 0.]) if nth < 1e-9 else th / nth
    ga = rel / np.linalg.norm(rel)
    return th, ga


# --------------------------------------------------------------------------------------
# A7 -- velocity transform
# --------------------------------------------------------------------------------------

def velocity_transform_table(R, v_world):
    """velocity_transform_batch.py:100-101 / batch_correct_velocity.py:38-45: R @ v per row."""
    R = np.asarray(R, float).reshape(-1, 3, 3); v_world = np.asarray(v_world, float)
    return np.einsum("tij,tj->ti", R, v_world)


def compute_rotation_kabsch(P, Q):
    """velocity_transform_batch.py:8-19 (== velocity_transform.py:42-54), statement by statement."""
    centroid_P = P.mean(axis=0)
    centroid_Q = Q.mean(axis=0)
    P_centered = P - centroid_P
    Q_centered = Q - centroid_Q
    H = P_centered.T @ Q_centered
    U, _, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        Vt[-1, :] *= -1
        R = Vt.T @ U.T
    return R


def kabsch_velocity_transform(original_points, corrected_points, rob_speed, batch_gates=True):
    """Per-frame loop of velocity_transform_batch.py:71-107 (batch_gates) / velocity_transform.py:60-80."""
    out, Rs = [], []
    nan3 = [np.nan] * 3
    for t in range(len(rob_speed)):
        P = original_points[t]; Q = corrected_points[t]
        bad = not (np.isfinite(P).all() and np.isfinite(Q).all())
        bad = bad or P.shape[0] < 3
        bad = bad or (batch_gates and np.linalg.norm(P - Q) < 1e-6)
        if bad:
            out.append(nan3); Rs.append(np.full((3, 3), np.nan)); continue
        R = compute_rotation_kabsch(P, Q)
        if batch_gates and (not np.allclose(R.T @ R, np.eye(3), atol=1e-2) or not np.isclose(np.linalg.det(R), 1.0, atol=1e-2)):
            out.append(nan3); Rs.append(np.full((3, 3), np.nan)); continue
        out.append(R @ rob_speed[t]); Rs.append(R)
    return np.array(out), np.array(Rs)


def velocity_transform_compose(v_world, rel, theta, gamma):
    """Build-defined world->catenary-frame rotation from (theta, gamma):
    v_cat = R_theta(+theta) R_gamma(-gamma) v, the inverse of the augmentation of
    main_fun.py:96-109 (points are rotated by -theta then +gamma)."""
    th_axis, ga_axis = theta_gamma_axes(rel)
    return rodrigues_rotation(rodrigues_rotation(v_world, ga_axis, -gamma), th_axis, theta)


# --------------------------------------------------------------------------------------
# A8 -- closed-loop rollout, cost, arg-min (build-defined)
# --------------------------------------------------------------------------------------

@dataclass
class MPCConfig:
    N: int = 20
    dt: float = 1.0 / 60.0
    v_scale: float = 1e-3            # V1 is mm/s, P1 is m (main_fun.py:815 divides V by 1000)
    L: float = 3.0                   # test_cluster.py:22
    cable_wet_weight: float = 1.521  # test_cluster.py:23
    c_lo: float = C_LO_DEFAULT
    c_hi: float = C_HI_DEFAULT
    n_shape_pts: int = 16
    up: float = 1.0                  # ENU
    vt_mode: int = 1                 # 0 none, 1 compose from (theta,gamma), 2 table
    prev_mode: int = 0               # 0 interpolate delay slots (reference midpoint), 1 hold
    integrator: int = 0              # 0 rk4, 1 euler
    w_theta: float = 1.0
    w_gamma: float = 1.0
    w_u: float = 1e-6
    w_T: float = 1e-2
    w_taut: float = 1e3
    rho_taut: float = 0.98
    w_floor: float = 10.0
    z_floor: float = -1.2
    theta_ref: float = 0.0
    gamma_ref: float = 0.0
    U_ref: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    feature_map: int = 0             # 0: 18 scaled slots (simply.py:15-41); 1: 17 unscaled slots (simulate_rk4_theta_gamma.py:12-42); 2: features_dd, second order (rollout_vec_dd)


@dataclass
class MPCState:
    P0: np.ndarray
    P1: np.ndarray
    V1: np.ndarray
    A1: np.ndarray
    theta: float
    gamma: float
    theta_prev: float
    gamma_prev: float

    def as_array(self):
        return np.concatenate([np.asarray(self.P0, float), np.asarray(self.P1, float),
                               np.asarray(self.V1, float), np.asarray(self.A1, float),
                               [self.theta, self.gamma, self.theta_prev, self.gamma_prev]])

    @staticmethod
    def from_array(a):
        a = np.asarray(a, float)
        return MPCState(a[0:3].copy(), a[3:6].copy(), a[6:9].copy(), a[9:12].copy(),
                        float(a[12]), float(a[13]), float(a[14]), float(a[15]))


@dataclass
class DynamicsModel:
    mean: np.ndarray
    scale: np.ndarray
    f_theta: SymbolicModel
    f_gamma: SymbolicModel


def _exo_features_scalar(P0, P, V, A, mean, scale, fmap=0):
    """simply.py:25-31 for one row; returns the 14 scaled exogenous slots (generation 2,
    simulate_rk4_theta_gamma.py:25-38: slot 12 unused, slot 13 = UNCLIPPED angle_proj scaled as x16)."""
    rel = P - P0
    nr = np.linalg.norm(rel)
    unit_rel = rel / (nr + 1e-8)
    tension = np.clip(nr, 1e-5, 10)
    angle_proj = np.dot(V, unit_rel) / (np.linalg.norm(V) + 1e-8)
    if fmap == 1:
        x = np.concatenate([P, V, A, unit_rel])
        return np.concatenate([(x - mean[:12]) / scale[:12], [0.0, (angle_proj - mean[16]) / scale[16]]])
    x = np.concatenate([P, V, A, unit_rel, [tension, np.clip(angle_proj, -1, 1)]])
    return (x - mean[:14]) / scale[:14]


def rollout_scalar(cfg: MPCConfig, model: DynamicsModel, state: MPCState, U, Rtab=None):
    """Reference-style scalar rollout: Python ``for k: for n:``, one-row model.predict
    per stage (simulate_rk4_theta_gamma.py:54-67), scipy brentq per node
    (main_fun.py:421-431) and per-point Rodrigues loops (main_fun.py:96-109).

    Returns (J(K,), traj(K,N+1,2), aux dict)."""
    U = np.asarray(U, float)
    K, N, _ = U.shape
    mean, scale = model.mean, model.scale
    h = cfg.dt
    cat = Catenary(cfg.L, "ENU" if cfg.up > 0 else "NED", cfg.n_shape_pts, cfg.c_lo, cfg.c_hi)
    J = np.zeros(K); traj = np.zeros((K, N + 1, 2))
    Tn = np.zeros((K, N)); Zn = np.zeros((K, N)); Cn = np.zeros((K, N))
    Uref = np.asarray(cfg.U_ref, float)
    P0 = np.asarray(state.P0, float)

    def f(xrow):
        X = xrow.reshape(1, -1)
        return model.f_theta.predict(X)[0], model.f_gamma.predict(X)[0]

    for k in range(K):
        P = np.asarray(state.P1, float).copy()
        V = np.asarray(state.V1, float).copy()
        A = np.asarray(state.A1, float).copy()
        th, ga, thm, gam = state.theta, state.gamma, state.theta_prev, state.gamma_prev
        traj[k, 0] = (th, ga)
        xs_n = _exo_features_scalar(P0, P, V, A, mean, scale, cfg.feature_map)
        Jk = 0.0
        for n in range(N):
            Uw = U[k, n]
            if cfg.vt_mode == 0:
                Vn = Uw.copy()
            elif cfg.vt_mode == 1:
                Vn = velocity_transform_compose(Uw, P - P0, th, ga)
            else:
                Vn = np.asarray(Rtab[n], float).reshape(3, 3) @ Uw
            Pn = P + (cfg.v_scale * h) * Uw
            An = (Vn - V) / h
            xs_n1 = _exo_features_scalar(P0, Pn, Vn, An, mean, scale, cfg.feature_map)

            def stage(yth, yga, c):
                if c == 0.0:
                    exo = xs_n
                elif c == 1.0:
                    exo = xs_n1
                else:
                    exo = (xs_n + xs_n1) / 2                      # :62 feature midpoint
                if cfg.feature_map == 1:        # simulate_rk4_theta_gamma.py:40
                    row = np.concatenate([exo[:12], [(yth - mean[12]) / scale[12], (yga - mean[13]) / scale[13],
                                                     (np.cos(yth) - mean[14]) / scale[14],
                                                     (np.sin(yga) - mean[15]) / scale[15], exo[13]]])
                    return f(row)
                s16a = (thm - mean[16]) / scale[16]; s16b = (th - mean[16]) / scale[16]
                s17a = (gam - mean[17]) / scale[17]; s17b = (ga - mean[17]) / scale[17]
                if cfg.prev_mode == 1 or c == 0.0:
                    p16, p17 = s16a, s17a
                elif c == 1.0:
                    p16, p17 = s16b, s17b
                else:
                    p16, p17 = (s16a + s16b) / 2, (s17a + s17b) / 2
                row = np.concatenate([exo, [(yth - mean[14]) / scale[14],
                                            (yga - mean[15]) / scale[15], p16, p17]])
                return f(row)

            k1 = stage(th, ga, 0.0)
            if cfg.integrator == 1:
                th_n = th + k1[0] * h                              # main_fun.py:761
                ga_n = ga + k1[1] * h
            else:
                k2 = stage(th + 0.5 * h * k1[0], ga + 0.5 * h * k1[1], 0.5)
                k3 = stage(th + 0.5 * h * k2[0], ga + 0.5 * h * k2[1], 0.5)
                k4 = stage(th + h * k3[0], ga + h * k3[1], 1.0)
                th_n = th + (h / 6) * (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0])   # :66
                ga_n = ga + (h / 6) * (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1])
            thm, gam = th, ga
            th, ga = th_n, ga_n
            P, V, A = Pn, Vn, An
            xs_n = xs_n1
            traj[k, n + 1] = (th, ga)

            rel = P - P0
            l = np.linalg.norm(rel[:2]); dH = cfg.up * rel[2]     # main_fun.py:292-293
            d = np.linalg.norm(rel)
            C = solve_catenary_scalar(l, dH, cfg.L, cfg.c_lo, cfg.c_hi)
            T = float(cable_tension(l, C, cfg.L, cfg.cable_wet_weight))
            with np.errstate(all="ignore"):
                final = transform_catenary(P0, P, cat, th, ga)[3]
                z_low = cfg.up * np.min(cfg.up * final[:, 2])
                c_n = (cfg.w_theta * (th - cfg.theta_ref) ** 2
                       + cfg.w_gamma * (ga - cfg.gamma_ref) ** 2
                       + cfg.w_u * float(np.sum((Uw - Uref) ** 2))
                       + cfg.w_T * T
                       + cfg.w_taut * max(0.0, d - cfg.rho_taut * cfg.L) ** 2
                       + cfg.w_floor * max(0.0, cfg.up * (cfg.z_floor - z_low)) ** 2)
            Jk = Jk + c_n
            Tn[k, n] = T; Zn[k, n] = z_low; Cn[k, n] = C
        J[k] = Jk if not np.isnan(Jk) else np.inf
    return J, traj, {"T": Tn, "z_low": Zn, "C": Cn}


# ---- K-vectorised flavour --------------------------------------------------------------

def _rod_vec(v, axis, ang):
    """Row-wise Rodrigues (axis re-normalised like main_fun.py:30)."""
    axis = axis / np.linalg.norm(axis, axis=-1, keepdims=True)
    c = np.cos(ang)[..., None]; s = np.sin(ang)[..., None]
    return v * c + np.cross(axis, v) * s + axis * np.sum(axis * v, axis=-1, keepdims=True) * (1 - c)


def _axes_vec(rel):
    xy = rel.copy(); xy[..., 2] = 0
    nxy = np.linalg.norm(xy, axis=-1, keepdims=True)
    deg = nxy < 1e-9
    with np.errstate(all="ignore"):
        xy = np.where(deg, np.array([1., 0., 0.]), xy / nxy)
    th = np.cross(xy, np.array([0., 0., 1.]))
    nth = np.linalg.norm(th, axis=-1, keepdims=True)
    with np.errstate(all="ignore"):
        th = np.where(nth < 1e-9, np.array([0., 1., 0.]), th / nth)
        ga = rel / np.linalg.norm(rel, axis=-1, keepdims=True)
    return th, ga


def _exo_features_vec(P0, P, V, A, mean, scale, fmap=0):
    rel = P - P0
    nr = np.linalg.norm(rel, axis=1, keepdims=True)
    unit_rel = rel / (nr + 1e-8)
    tension = np.clip(nr, 1e-5, 10)
    with np.errstate(all="ignore"):
        angle_proj = np.sum(V * unit_rel, axis=1, keepdims=True) / (np.linalg.norm(V, axis=1, keepdims=True) + 1e-8)
    if fmap == 1:
        x = np.hstack([P, V, A, unit_rel])
        return np.hstack([(x - mean[:12]) / scale[:12], np.zeros_like(nr), (angle_proj - mean[16]) / scale[16]])
    x = np.hstack([P, V, A, unit_rel, tension, np.clip(angle_proj, -1, 1)])
    return (x - mean[:14]) / scale[:14]


def augmented_lowest_z_vec(P0, P, th, ga, L, M, up, c_lo, c_hi):
    """z of the lowest sample of transform_catenary(P0, P, Catenary(L), th, ga)[3], per row."""
    rel = P - P0
    th_axis, ga_axis = _axes_vec(rel)
    Bp_rel = _rod_vec(rel, th_axis, th)                        # main_fun.py:92
    lp = np.sqrt(Bp_rel[:, 0] ** 2 + Bp_rel[:, 1] ** 2)
    dHp = up * Bp_rel[:, 2]
    Cp = solve_catenary_vec(lp, dHp, L, c_lo, c_hi)
    valid = np.isfinite(Cp)
    Cs = np.where(valid, Cp, 1.0)
    with np.errstate(all="ignore"):
        x0 = 0.5 * lp - np.arctanh(np.where(valid, dHp / L, 0.0)) / Cs
        ch0 = np.cosh(Cs * x0)
    best = np.full(P.shape[0], np.inf)
    for j in range(M):
        t = j / (M - 1)
        with np.errstate(all="ignore"):
            up_j = (np.cosh(Cs * (lp * t - x0)) - ch0) / Cs
        q = np.stack([t * Bp_rel[:, 0], t * Bp_rel[:, 1], up * up_j], axis=1)   # pt - A
        q = _rod_vec(q, th_axis, -th)                           # :96-99
        q = _rod_vec(q, ga_axis, ga)                            # :106-109
        zj = P0[2] + q[:, 2]
        best = np.where(valid, np.minimum(best, up * zj), best)
    # straight-segment fallback (main_fun.py:67-69): points [A, B']
    qa = np.zeros_like(rel)
    qb = _rod_vec(_rod_vec(Bp_rel, th_axis, -th), ga_axis, ga)
    qa = _rod_vec(_rod_vec(qa, th_axis, -th), ga_axis, ga)
    seg = np.minimum(up * (P0[2] + qa[:, 2]), up * (P0[2] + qb[:, 2]))
    best = np.where(valid, best, seg)
    return up * best


def rollout_vec(cfg: MPCConfig, model: DynamicsModel, state: MPCState, U, Rtab=None):
    """K-vectorised NumPy flavour of ``rollout_scalar`` (same arithmetic per candidate)."""
    U = np.asarray(U, float)
    K, N, _ = U.shape
    mean, scale = model.mean, model.scale
    h = cfg.dt
    P0 = np.asarray(state.P0, float)
    P = np.tile(np.asarray(state.P1, float), (K, 1))
    V = np.tile(np.asarray(state.V1, float), (K, 1))
    A = np.tile(np.asarray(state.A1, float), (K, 1))
    th = np.full(K, float(state.theta)); ga = np.full(K, float(state.gamma))
    thm = np.full(K, float(state.theta_prev)); gam = np.full(K, float(state.gamma_prev))
    Uref = np.asarray(cfg.U_ref, float)
    traj = np.zeros((K, N + 1, 2)); traj[:, 0, 0] = th; traj[:, 0, 1] = ga
    J = np.zeros(K)
    Tn = np.zeros((K, N)); Zn = np.zeros((K, N)); Cn = np.zeros((K, N))
    xs_n = _exo_features_vec(P0, P, V, A, mean, scale, cfg.feature_map)
    if cfg.feature_map == 1:
        s16a = s16b = s17a = s17b = None

    def f(cols):
        a = np.broadcast_to(np.asarray(model.f_theta(cols), float), (K,))
        b = np.broadcast_to(np.asarray(model.f_gamma(cols), float), (K,))
        return a, b

    with np.errstate(all="ignore"):
        for n in range(N):
            Uw = U[:, n, :]
            if cfg.vt_mode == 0:
                Vn = Uw.copy()
            elif cfg.vt_mode == 1:
                tha, gaa = _axes_vec(P - P0)
                Vn = _rod_vec(_rod_vec(Uw, gaa, -ga), tha, th)
            else:
                Vn = Uw @ np.asarray(Rtab[n], float).reshape(3, 3).T
            Pn = P + (cfg.v_scale * h) * Uw
            An = (Vn - V) / h
            xs_n1 = _exo_features_vec(P0, Pn, Vn, An, mean, scale, cfg.feature_map)
            if cfg.feature_map != 1:
                s16a = (thm - mean[16]) / scale[16]; s16b = (th - mean[16]) / scale[16]
                s17a = (gam - mean[17]) / scale[17]; s17b = (ga - mean[17]) / scale[17]

            def stage(yth, yga, c):
                if c == 0.0:
                    exo = xs_n
                elif c == 1.0:
                    exo = xs_n1
                else:
                    exo = (xs_n + xs_n1) / 2
                if cfg.feature_map == 1:
                    cols = [exo[:, i] for i in range(12)]
                    cols += [(yth - mean[12]) / scale[12], (yga - mean[13]) / scale[13],
                             (np.cos(yth) - mean[14]) / scale[14], (np.sin(yga) - mean[15]) / scale[15], exo[:, 13]]
                    return f(cols)
                if cfg.prev_mode == 1 or c == 0.0:
                    p16, p17 = s16a, s17a
                elif c == 1.0:
                    p16, p17 = s16b, s17b
                else:
                    p16, p17 = (s16a + s16b) / 2, (s17a + s17b) / 2
                cols = [exo[:, i] for i in range(14)]
                cols += [(yth - mean[14]) / scale[14], (yga - mean[15]) / scale[15], p16, p17]
                return f(cols)

            k1 = stage(th, ga, 0.0)
            if cfg.integrator == 1:
                th_n = th + k1[0] * h
                ga_n = ga + k1[1] * h
            else:
                k2 = stage(th + 0.5 * h * k1[0], ga + 0.5 * h * k1[1], 0.5)
                k3 = stage(th + 0.5 * h * k2[0], ga + 0.5 * h * k2[1], 0.5)
                k4 = stage(th + h * k3[0], ga + h * k3[1], 1.0)
                th_n = th + (h / 6) * (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0])
                ga_n = ga + (h / 6) * (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1])
            thm, gam = th, ga
            th, ga = th_n, ga_n
            P, V, A = Pn, Vn, An
            xs_n = xs_n1
            traj[:, n + 1, 0] = th; traj[:, n + 1, 1] = ga

            rel = P - P0
            l = np.sqrt(rel[:, 0] ** 2 + rel[:, 1] ** 2); dH = cfg.up * rel[:, 2]
            d = np.linalg.norm(rel, axis=1)
            C = solve_catenary_vec(l, dH, cfg.L, cfg.c_lo, cfg.c_hi)
            T = cable_tension(l, C, cfg.L, cfg.cable_wet_weight)
            z_low = augmented_lowest_z_vec(P0, P, th, ga, cfg.L, cfg.n_shape_pts, cfg.up,
                                           cfg.c_lo, cfg.c_hi)
            c_n = (cfg.w_theta * (th - cfg.theta_ref) ** 2
                   + cfg.w_gamma * (ga - cfg.gamma_ref) ** 2
                   + cfg.w_u * np.sum((Uw - Uref) ** 2, axis=1)
                   + cfg.w_T * T
                   + cfg.w_taut * np.maximum(0.0, d - cfg.rho_taut * cfg.L) ** 2
                   + cfg.w_floor * np.maximum(0.0, cfg.up * (cfg.z_floor - z_low)) ** 2)
            J = J + c_n
            Tn[:, n] = T; Zn[:, n] = z_low; Cn[:, n] = C
    J = np.where(np.isnan(J), np.inf, J)
    return J, traj, {"T": Tn, "z_low": Zn, "C": Cn}


# --------------------------------------------------------------------------------------
# N2 -- second-order models: features_dd (main_fun.py:811-871) and the closed-loop rollout on
#       the state (theta, gamma, dtheta, dgamma)
# --------------------------------------------------------------------------------------

DD_NAMES = ("theta", "gama", "dtheta", "dgamma", "v_sway", "v_surge", "a_sway", "a_surge",
            "V_x", "V_y", "V_z", "a_x", "a_y", "a_z")          # dd_cluster.py:160-168


def features_dd(P0_mm, P1_mm, V_mm, time, theta_raw, gamma_raw):
    """main_fun.py:811-871 on arrays: returns (features(T,14), targets(T,2))."""
    from scipy.signal import savgol_filter
    P0 = np.asarray(P0_mm, float) / 1000; P1 = np.asarray(P1_mm, float) / 1000; V1 = np.asarray(V_mm, float) / 1000   # :813-815
    t = np.asarray(time, float)
    a = np.stack([np.gradient(V1[:, j], t) for j in range(3)], axis=1)                       # :825-827
    th = savgol_filter(np.asarray(theta_raw, float), window_length=11, polyorder=3)          # :830-831
    ga = savgol_filter(np.asarray(gamma_raw, float), window_length=11, polyorder=3)
    dth = np.gradient(th, t); dga = np.gradient(ga, t)                                       # :833-834
    ddth = np.gradient(dth, t); ddga = np.gradient(dga, t)                                   # :835-836
    rel = P1 - P0
    unit = rel / (np.linalg.norm(rel, axis=1, keepdims=True) + 1e-8)                         # :841
    v_surge = np.sum(V1 * unit, axis=1)                                                      # :842
    v_sway = np.linalg.norm(np.cross(V1, unit), axis=1)                                      # :843
    a_surge = np.gradient(v_surge, t); a_sway = np.gradient(v_sway, t)                       # :846-847
    feats = np.stack([th, ga, dth, dga, v_sway, v_surge, a_sway, a_surge,
                      V1[:, 0], V1[:, 1], V1[:, 2], a[:, 0], a[:, 1], a[:, 2]], axis=1)       # :849-864
    return feats, np.stack([ddth, ddga], axis=1)


def _dd_surge_sway(P0, P, Vm):
    rel = P - P0
    unit = rel / (np.linalg.norm(rel, axis=-1, keepdims=True) + 1e-8)
    return np.linalg.norm(np.cross(Vm, unit), axis=-1), np.sum(Vm * unit, axis=-1)            # sway, surge


def rollout_vec_dd(cfg: MPCConfig, model: DynamicsModel, state: MPCState, U, Rtab=None):
    """Closed-loop rollout of a SECOND-order model pair (ddtheta, ddgamma) = f(scaled features_dd row),
    build-defined like the first-order one:  y = (theta, gamma, dtheta, dgamma),
    y' = (dtheta, dgamma, f_theta(x), f_gamma(x)); classic RK4 (integrator 0) or the reference's explicit
    double Euler (test_cluster.py:110-129, integrator 1).  Row x: slots 0-3 the stage state, slots 4-13
    exogenous (surge/sway speeds and their first differences, V and A in m/s), interpolated between
    nodes n and n+1 at the midpoint stages (simulate_rk4_theta_gamma.py:62).  a_sway/a_surge at node 0
    take np.gradient's edge rule (= the first difference of nodes 0,1, main_fun.py:846-847).
    ``state.theta_prev / gamma_prev`` carry dtheta_0 / dgamma_0 in this map."""
    U = np.asarray(U, float)
    K, N, _ = U.shape
    mean, scale = np.asarray(model.mean, float), np.asarray(model.scale, float)
    h, vs = cfg.dt, cfg.v_scale
    P0 = np.asarray(state.P0, float)
    P = np.tile(np.asarray(state.P1, float), (K, 1))
    V = np.tile(np.asarray(state.V1, float), (K, 1))
    A = np.tile(np.asarray(state.A1, float), (K, 1))
    th = np.full(K, float(state.theta)); ga = np.full(K, float(state.gamma))
    dth = np.full(K, float(state.theta_prev)); dga = np.full(K, float(state.gamma_prev))
    Uref = np.asarray(cfg.U_ref, float)
    traj = np.zeros((K, N + 1, 2)); traj[:, 0, 0] = th; traj[:, 0, 1] = ga
    J = np.zeros(K)
    Tn = np.zeros((K, N)); Zn = np.zeros((K, N)); Cn = np.zeros((K, N))

    def f(cols):
        a = np.broadcast_to(np.asarray(model.f_theta(cols), float), (K,))
        b = np.broadcast_to(np.asarray(model.f_gamma(cols), float), (K,))
        return a, b

    def exo_row(sway, surge, a_sway, a_surge, Vm, Am):
        x = np.stack([sway, surge, a_sway, a_surge, Vm[:, 0], Vm[:, 1], Vm[:, 2], Am[:, 0], Am[:, 1], Am[:, 2]], axis=1)
        return (x - mean[4:14]) / scale[4:14]

    with np.errstate(all="ignore"):
        sway, surge = _dd_surge_sway(P0, P, vs * V)
        xs_n = None
        for n in range(N):
            Uw = U[:, n, :]
            if cfg.vt_mode == 0:
                Vn = Uw.copy()
            elif cfg.vt_mode == 1:
                tha, gaa = _axes_vec(P - P0)
                Vn = _rod_vec(_rod_vec(Uw, gaa, -ga), tha, th)
            else:
                Vn = Uw @ np.asarray(Rtab[n], float).reshape(3, 3).T
            Pn = P + (vs * h) * Uw
            An = (Vn - V) / h
            sway_n, surge_n = _dd_surge_sway(P0, Pn, vs * Vn)
            a_sway_n = (sway_n - sway) / h; a_surge_n = (surge_n - surge) / h
            if n == 0:
                xs_n = exo_row(sway, surge, a_sway_n, a_surge_n, vs * V, vs * A)              # edge rule at node 0
            xs_n1 = exo_row(sway_n, surge_n, a_sway_n, a_surge_n, vs * Vn, vs * An)

            def stage(y, c):
                exo = xs_n if c == 0.0 else xs_n1 if c == 1.0 else (xs_n + xs_n1) / 2
                cols = [(y[i] - mean[i]) / scale[i] for i in range(4)] + [exo[:, i] for i in range(10)]
                ft, fg = f(cols)
                return (y[2], y[3], ft, fg)

            y = (th, ga, dth, dga)
            k1 = stage(y, 0.0)
            if cfg.integrator == 1:
                yn = tuple(y[i] + k1[i] * h for i in range(4))                                # test_cluster.py:113-129
            else:
                k2 = stage(tuple(y[i] + 0.5 * h * k1[i] for i in range(4)), 0.5)
                k3 = stage(tuple(y[i] + 0.5 * h * k2[i] for i in range(4)), 0.5)
                k4 = stage(tuple(y[i] + h * k3[i] for i in range(4)), 1.0)
                yn = tuple(y[i] + (h / 6) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) for i in range(4))
            th, ga, dth, dga = yn
            P, V, A = Pn, Vn, An
            sway, surge = sway_n, surge_n
            xs_n = xs_n1
            traj[:, n + 1, 0] = th; traj[:, n + 1, 1] = ga

            rel = P - P0
            l = np.sqrt(rel[:, 0] ** 2 + rel[:, 1] ** 2); dH = cfg.up * rel[:, 2]
            d = np.linalg.norm(rel, axis=1)
            C = solve_catenary_vec(l, dH, cfg.L, cfg.c_lo, cfg.c_hi)
            T = cable_tension(l, C, cfg.L, cfg.cable_wet_weight)
            z_low = augmented_lowest_z_vec(P0, P, th, ga, cfg.L, cfg.n_shape_pts, cfg.up, cfg.c_lo, cfg.c_hi)
            c_n = (cfg.w_theta * (th - cfg.theta_ref) ** 2
                   + cfg.w_gamma * (ga - cfg.gamma_ref) ** 2
                   + cfg.w_u * np.sum((Uw - Uref) ** 2, axis=1)
                   + cfg.w_T * T
                   + cfg.w_taut * np.maximum(0.0, d - cfg.rho_taut * cfg.L) ** 2
                   + cfg.w_floor * np.maximum(0.0, cfg.up * (cfg.z_floor - z_low)) ** 2)
            J = J + c_n
            Tn[:, n] = T; Zn[:, n] = z_low; Cn[:, n] = C
    J = np.where(np.isnan(J), np.inf, J)
    return J, traj, {"T": Tn, "z_low": Zn, "C": Cn, "dtheta": dth, "dgamma": dga}


def rollout_scalar_dd(cfg: MPCConfig, model: DynamicsModel, state: MPCState, U, Rtab=None):
    """Reference-style scalar flavour of ``rollout_vec_dd``: Python loops, one-row ``predict`` per stage.
    Trajectories only (the node cost is the first-order rollout's, checked there)."""
    U = np.asarray(U, float)
    K, N, _ = U.shape
    mean, scale = np.asarray(model.mean, float), np.asarray(model.scale, float)
    h, vs = cfg.dt, cfg.v_scale
    P0 = np.asarray(state.P0, float)
    traj = np.zeros((K, N + 1, 2))
    for k in range(K):
        P = np.asarray(state.P1, float).copy(); V = np.asarray(state.V1, float).copy(); A = np.asarray(state.A1, float).copy()
        y = np.array([state.theta, state.gamma, state.theta_prev, state.gamma_prev], float)
        traj[k, 0] = y[:2]
        sway, surge = _dd_surge_sway(P0, P, vs * V)
        xs_n = None
        for n in range(N):
            Uw = U[k, n]
            if cfg.vt_mode == 0:
                Vn = Uw.copy()
            elif cfg.vt_mode == 1:
                Vn = velocity_transform_compose(Uw, P - P0, y[0], y[1])
            else:
                Vn = np.asarray(Rtab[n], float).reshape(3, 3) @ Uw
            Pn = P + (vs * h) * Uw
            An = (Vn - V) / h
            sway_n, surge_n = _dd_surge_sway(P0, Pn, vs * Vn)
            a_sw = (sway_n - sway) / h; a_su = (surge_n - surge) / h
            if n == 0:
                xs_n = (np.concatenate([[sway, surge, a_sw, a_su], vs * V, vs * A]) - mean[4:14]) / scale[4:14]
            xs_n1 = (np.concatenate([[sway_n, surge_n, a_sw, a_su], vs * Vn, vs * An]) - mean[4:14]) / scale[4:14]

            def stage(yy, c):
                exo = xs_n if c == 0.0 else xs_n1 if c == 1.0 else (xs_n + xs_n1) / 2
                row = np.concatenate([(yy - mean[:4]) / scale[:4], exo]).reshape(1, -1)
                return np.array([yy[2], yy[3], model.f_theta.predict(row)[0], model.f_gamma.predict(row)[0]])

            with np.errstate(all="ignore"):
                k1 = stage(y, 0.0)
                if cfg.integrator == 1:
                    y = y + h * k1
                else:
                    k2 = stage(y + 0.5 * h * k1, 0.5); k3 = stage(y + 0.5 * h * k2, 0.5); k4 = stage(y + h * k3, 1.0)
                    y = y + (h / 6) * (k1 + 2 * k2 + 2 * k3 + k4)
            P, V, A, sway, surge, xs_n = Pn, Vn, An, sway_n, surge_n, xs_n1
            traj[k, n + 1] = y[:2]
    return traj


def mpc_step(cfg, model, state, U, Rtab=None, flavour="vec"):
    """Build-defined ``step``: returns (u(3), traj(N+1,2), J*, k*) with np.argmin tie-break."""
    fn = rollout_vec if flavour == "vec" else rollout_scalar
    J, traj, _ = fn(cfg, model, state, U, Rtab)
    k = int(np.argmin(J))
    return np.asarray(U)[k, 0, :].copy(), traj[k].copy(), float(J[k]), k


def closed_loop(cfg, model, rows, pools, feedback, n_steps=None, Rtab=None, k_offset=0):
    """BASELINE config 5: an MPC step per sample of a trajectory table (build-defined -- the reference has no controller;
    its per-frame loop over recorded rows is catenary_from_data.py:40-50, the table comes from Rov_traj_gen.py:7-116).

    Plant rule = rovmpc_closed_loop_device / plant_update_kernel: step i starts from the measured row ``rows[i]`` (16
    doubles in rovmpc_state order: P0, P1, V1, A1, theta, gamma, theta_prev, gamma_prev).  With ``feedback``, from the second
    step on only the exogenous slots 0..11 come from the row; (theta, gamma) = the first predicted node of step i - 1's
    winner, and (theta_prev, gamma_prev) = the (theta, gamma) step i - 1 started from (np.roll, simply.py:35-38).  Step i
    rolls out candidate batch ``pools[i % len(pools)]`` (``rollout_vec``) and takes np.argmin (lowest index on ties).

    Returns dict(cost (T,), index (T,) + k_offset, u (T, 3), theta_gamma (T + 1, 2), states (T, 16), traj (T, N + 1, 2))."""
    rows = np.asarray(rows, dtype=np.float64)
    T = len(rows) if n_steps is None else int(n_steps)
    cost = np.empty(T); index = np.empty(T, dtype=np.int64); u = np.empty((T, 3))
    states = np.empty((T, 16)); trajs = np.empty((T, cfg.N + 1, 2))
    st = rows[0].copy()
    for i in range(T):
        if feedback and i > 0:
            th, ga = st[12], st[13]
            st = st.copy()
            st[0:12] = rows[i][0:12]
            st[14], st[15] = th, ga
            st[12], st[13] = trajs[i - 1, 1]
        else:
            st = rows[i].copy()
        U = np.asarray(pools[i % len(pools)], dtype=np.float64)
        J, traj, _ = rollout_vec(cfg, model, MPCState.from_array(st), U, Rtab)
        k = int(np.argmin(J))
        cost[i] = J[k]; index[i] = k + k_offset; u[i] = U[k, 0]; states[i] = st; trajs[i] = traj[k]
    tg = np.vstack([trajs[0, 0], trajs[:, 1]])
    return {"cost": cost, "index": index, "u": u, "theta_gamma": tg, "states": states, "traj": trajs}


# --------------------------------------------------------------------------------------
# A9 -- ROV trajectory generator (Rov_traj_gen.py:7-116)
# --------------------------------------------------------------------------------------

def rov_trajectories(exp_case: int, n_steps: int = 100, total_time: float = 10.0,
                     separation: float = 1.0, seed: Optional[int] = None):
    """Returns (time(n_steps,), trajectory_0(12,n_steps), trajectory_1(12,n_steps))."""
    time = np.linspace(0, total_time, n_steps)                               # :9
    t0 = np.zeros((12, n_steps)); t1 = np.zeros((12, n_steps))               # :13-14
    rng = np.random.default_rng(seed)
    if exp_case == 1:
        t0[0] = 0.03 * time; t1[0] = 0.03 * time; t1[1] = separation; t0[6] = 0.03; t1[6] = 0.03
    elif exp_case == 2:
        t0[0] = 0.03 * time; t1[0] = 0.06 * time; t1[1] = separation; t0[6] = 0.03; t1[6] = 0.06
    elif exp_case == 3:
        t0[0] = 0.03 * time; t1[0] = -0.03 * time; t1[1] = separation; t0[6] = 0.03; t1[6] = -0.03
    elif exp_case == 4:
        t0[0] = 0; t1[0] = 0.05 * time; t1[1] = separation; t1[6] = 0.5
    elif exp_case == 5:
        t0[0] = 0.03 * time; t1[0] = 0.03 * time; t1[1] = separation
        t0[2] = 0.5; t1[2] = np.linspace(0.5, 1.0, n_steps); t0[6] = 0.03; t1[6] = 0.03
    elif exp_case == 6:
        t0[0] = 0.03 * time; t1[0] = 0.06 * time; t1[1] = separation
        t0[2] = 0.5; t1[2] = np.linspace(0.5, 1.0, n_steps); t0[6] = 0.03; t1[6] = 0.06
    elif exp_case == 7:
        t1[1] = separation; t0[2] = 0.5; t1[2] = np.linspace(0.5, 1.0, n_steps)
        t0[6] = 0; t1[6] = 0.05
    elif exp_case == 8:
        t0[0] = 0.05 * time; t1[0] = 0.05 * time
        t0[1] = 0.05 * np.sin(2 * np.pi * time); t1[1] = separation + 0.05 * np.sin(2 * np.pi * time)
        t0[6] = 0.05 * np.cos(2 * np.pi * time / total_time)
        t1[6] = 0.05 * np.cos(2 * np.pi * time / total_time)
    elif exp_case == 9:     # :81-86 uses unseeded np.random; seeded here
        t0[0] = rng.choice([-0.1, 0.1], n_steps); t1[0] = 0.05 * time; t1[1] = separation
        t0[6] = rng.choice([-0.03, 0.03], n_steps)
    elif exp_case == 10:    # :87-92
        t0[0] = rng.choice([-0.1, 0.1], n_steps); t1[0] = rng.choice([-0.1, 0.1], n_steps)
        t1[1] = separation
        t0[6] = rng.choice([-0.03, 0.03], n_steps); t1[6] = rng.choice([-0.03, 0.03], n_steps)
    elif exp_case == 11:
        t0[0] = 0.05 * time; t1[0] = 0.05 * time; t1[1] = separation
        t0[1] = 0.2 * np.sin(2 * np.pi * time); t1[6] = 0.03
    elif exp_case == 12:
        t0[0] = 0.4 * np.cos(2 * np.pi * time / total_time)
        t0[1] = 0.4 * np.sin(2 * np.pi * time / total_time)
        t1[0] = 0.1 * np.cos(2 * np.pi * time / total_time)
        t1[1] = 0.1 * np.sin(2 * np.pi * time / total_time)
    elif exp_case == 13:
        t1[1] = separation; t0[0] = 0.06 * time; t1[0] = 0.06 * time; t0[6] = 0.06; t1[6] = 0.06
    elif exp_case == 14:
        t1[1] = separation
    else:
        raise ValueError("exp_case must be 1..14")
    return time, t0, t1


def rov_trajectory_csv_rows(t0, t1):
    """Rov_traj_gen.py:131-139 formatting (``%.3f``)."""
    rows = []
    for s0, s1 in zip(t0.T, t1.T):
        rows.append(",".join(f"{v:.3f}" for v in s0) + "," + ",".join(f"{v:.3f}" for v in s1))
    return rows


# ---- candidate sampler (build-defined: the reference has no MPC and no proposal law) ---------------------------------
# Restates csrc/util_kernels.h::sample_candidates_kernel: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random
# numbers: as easy as 1, 2, 3", SC'11 -- constants and round function as published) + Box-Muller.

def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over uint32 arrays c0..c3 (counter words); scalar key words k0, k1.  Returns the four output words."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    c0 = c0.astype(np.uint32); c1 = c1.astype(np.uint32); c2 = c2.astype(np.uint32); c3 = c3.astype(np.uint32)
    k0 = int(k0) & 0xFFFFFFFF; k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64); p1 = M1 * c2.astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ np.uint32(k0)
        n1 = (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ np.uint32(k1)
        n3 = (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF; k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def philox_normals(seed: int, step: int, n: int) -> np.ndarray:
    """n standard normals z_e, e = 0..n-1: block j = e // 4 = Philox(counter (j lo, j hi, step lo, step hi), key (seed lo,
    seed hi)); u_i = (x_i + 0.5) 2^-32; (z_4j, z_4j+1) = sqrt(-2 ln u0) (cos, sin)(2 pi u1), (z_4j+2, z_4j+3) from (u2, u3)."""
    nb = (n + 3) // 4
    j = np.arange(nb, dtype=np.uint64)
    x = philox4x32_10((j & np.uint64(0xFFFFFFFF)).astype(np.uint32), (j >> np.uint64(32)).astype(np.uint32),
                      np.full(nb, step & 0xFFFFFFFF, np.uint32), np.full(nb, (step >> 32) & 0xFFFFFFFF, np.uint32),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = [(xi.astype(np.float64) + 0.5) * 2.0 ** -32 for xi in x]
    z = np.empty((nb, 4))
    for h in range(2):
        r = np.sqrt(-2.0 * np.log(u[2 * h])); ang = 6.283185307179586 * u[2 * h + 1]
        z[:, 2 * h] = r * np.cos(ang); z[:, 2 * h + 1] = r * np.sin(ang)
    return z.reshape(-1)[:n]


def sample_candidates(seed: int, step: int, K: int, N: int, mean, std, prev_best=None) -> np.ndarray:
    """U[k, n, c] = mean[c] + std[c] z_e, e = (k N + n) 3 + c; candidate 0 = prev_best shifted by one step (warm start)."""
    z = philox_normals(seed, step, K * N * 3).reshape(K, N, 3)
    U = np.asarray(mean, float) + np.asarray(std, float) * z
    if prev_best is not None:
        U[0] = np.vstack([prev_best[1:], prev_best[-1:]])
    return U


# ---- Lagrangian evaluation (lagrangian_pipeline*.py, evaluate_lagrangian_on_test.py) -----------------------
# The reference's own route: sympy differentiation + lambdify.  sympy is imported lazily (tests / golden generation only).

def el_equations(lagrangian: str):
    """(EOM_theta, EOM_gamma) as sympy expressions over (th, ga, dth, dga, ddth, ddga): lagrangian_pipeline_old.py:60-83
    (x0..x3 of the equation text are theta, gamma, dtheta, dgamma: the `replacements` of :62)."""
    import sympy as sp
    # real=True only matters for Abs (its derivative is sign(x) for real x; the reference's plain symbols would leave an
    # unevaluated Derivative(re(..)) there); every other operator differentiates identically
    th, ga, dth, dga, ddth, ddga = sp.symbols("th ga dth dga ddth ddga", real=True)
    L = sp.sympify(lagrangian, locals={"x0": th, "x1": ga, "x2": dth, "x3": dga, "square": lambda v: v ** 2, "neg": lambda v: -v})
    out = []
    for q, dq in ((th, dth), (ga, dga)):
        p = sp.diff(L, dq)                                                       # :66 / :76
        d_p = sp.diff(p, th) * dth + sp.diff(p, ga) * dga + sp.diff(p, dth) * ddth + sp.diff(p, dga) * ddga   # :68-71
        out.append(d_p - sp.diff(L, q))                                          # :72
    return out[0], out[1], (th, ga, dth, dga, ddth, ddga)


def el_residuals(lagrangian: str, theta, gamma, dtheta, dgamma, ddtheta, ddgamma):
    """`evaluate` of lagrangian_pipeline_old.py:85-90: the two residual series."""
    import sympy as sp
    e_th, e_ga, syms = el_equations(lagrangian)
    args = [np.asarray(v, float) for v in (theta, gamma, dtheta, dgamma, ddtheta, ddgamma)]
    f_th = sp.lambdify(syms, e_th, modules="numpy"); f_ga = sp.lambdify(syms, e_ga, modules="numpy")
    z = np.zeros_like(args[0])
    return np.asarray(f_th(*args), float) + z, np.asarray(f_ga(*args), float) + z


def lagrangian_accelerations(lagrangian: str):
    """(ddtheta(th, ga, dth, dga), ddgamma(...)) callables: sp.solve(EOM, ddq)[0] of lagrangian_pipeline.py:146-171."""
    import sympy as sp
    e_th, e_ga, (th, ga, dth, dga, ddth, ddga) = el_equations(lagrangian)
    s_th = sp.solve(e_th, ddth)[0]; s_ga = sp.solve(e_ga, ddga)[0]
    return (sp.lambdify([th, ga, dth, dga], s_th, modules="numpy"), sp.lambdify([th, ga, dth, dga], s_ga, modules="numpy"))


def lagrangian_rollout(acc_theta, acc_gamma, time, theta0, gamma0, vtheta0, vgamma0):
    """evaluate_lagrangian_on_test.py:59-68, statement by statement."""
    T = len(time)
    theta_est = np.zeros(T); gamma_est = np.zeros(T); vtheta = np.zeros(T); vgamma = np.zeros(T)
    theta_est[0], gamma_est[0], vtheta[0], vgamma[0] = theta0, gamma0, vtheta0, vgamma0
    for i in range(1, T):
        dt = time[i] - time[i - 1]
        a_th = acc_theta(theta_est[i - 1], gamma_est[i - 1], vtheta[i - 1], vgamma[i - 1])
        a_ga = acc_gamma(theta_est[i - 1], gamma_est[i - 1], vtheta[i - 1], vgamma[i - 1])
        vtheta[i] = vtheta[i - 1] + a_th * dt
        theta_est[i] = theta_est[i - 1] + vtheta[i - 1] * dt
        vgamma[i] = vgamma[i - 1] + a_ga * dt
        gamma_est[i] = gamma_est[i - 1] + vgamma[i - 1] * dt
    return theta_est, gamma_est, vtheta, vgamma
