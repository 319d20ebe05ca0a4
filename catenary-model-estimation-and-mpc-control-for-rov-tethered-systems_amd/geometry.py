"""Drop-in mirrors of the reference's geometry callables, computed by librovmpc on the GPU.

    rodrigues_rotation(vector, axis, angle_rad)                      main_fun.py:18-35
    transform_catenary(point_A, point_B, catenary_fn, theta, gamma)  main_fun.py:38-111
    solve_catenary(l, delta_H, L)                                    main_fun.py:418-431
    Catenary(length=3., reference_frame='ENU')(a, b)                 catenary.py:10,25-29
    compute_catenary_3D(p0, p1, rope_length, num_points), Catenary3D models/catenary_3d.py:5-39
    lowest_point(points)                                             fully_augmented_catenary.py:21-22
    velocity_transform(R, v)                                         velocity_transform_batch.py:100-101
    compute_rotation_kabsch(P, Q), kabsch_velocity_transform(...)    velocity_transform_batch.py:8-19, 71-107

Same names, argument order and return shapes as the reference; array arguments may also be
batched (leading dimension) where the reference loops row by row.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from . import _lib
from .engine import Engine, MPCConfig, default_engine

_engines = {}


def _engine_for(frame: str, c_lo: float, c_hi: float) -> Engine:
    key = (frame, c_lo, c_hi)
    if key == ("ENU", 1e-6, 10.0):
        return default_engine()
    if key not in _engines:
        _engines[key] = Engine(MPCConfig(N=1, K=1, frame=frame, c_lo=c_lo, c_hi=c_hi))
    return _engines[key]


def rodrigues_rotation(vector, axis, angle_rad):
    v = np.asarray(vector, dtype=np.float64)
    out = default_engine().rodrigues(v.reshape(-1, 3), axis, angle_rad)
    return out.reshape(v.shape)


def solve_catenary(l, delta_H, L):
    """NaN where scipy's brentq would raise on the bracket [1e-6, 10] (main_fun.py:424-428).
    L must be a scalar (the reference always passes one)."""
    L = np.asarray(L, dtype=np.float64)
    if L.size != 1:
        raise ValueError("L must be a scalar")
    return default_engine().solve_catenary(l, delta_H, float(L.reshape(-1)[0]))


def cable_tension(l, delta_H, L, cable_wet_weight=1.521):
    """(C, T) with the static fallback T = w l / 2 where C is NaN (main_fun.py:302-305)."""
    eng = default_engine()
    if cable_wet_weight != eng.cfg.cable_wet_weight:
        key = ("w", float(cable_wet_weight))
        if key not in _engines:
            _engines[key] = Engine(MPCConfig(N=1, K=1, cable_wet_weight=float(cable_wet_weight)))
        eng = _engines[key]
    return eng.solve_catenary(l, delta_H, float(L), with_tension=True)


class Catenary:
    """``Catenary(length=3., reference_frame='ENU')``; ``catenary(a, b)`` returns a 4-tuple whose
    ``[3]`` is the ``(M,3)`` point array from a to b, or ``None`` when no catenary of that
    length spans the points inside the solver bracket (callers then draw the straight
    segment, catenary.py:25-29 / main_fun.py:67-69).  ``[0:3]`` = (C, sag, x_low).

    The original class lives in the un-vendored ``pympc`` package (absent from the reference
    snapshot); this one implements the in-repo catenary physics -- see DESIGN.md, "parity
    unpinned"."""

    def __init__(self, length: float = 3.0, reference_frame: str = "ENU", n_points: int = 32,
                 c_lo: float = 1e-6, c_hi: float = 10.0):
        if reference_frame not in ("ENU", "NED"):
            raise ValueError("reference_frame must be 'ENU' or 'NED'")
        if n_points < 2:
            raise ValueError("n_points must be >= 2")
        self.length = float(length)
        self.reference_frame = reference_frame
        self.n_points = int(n_points)
        self.c_lo, self.c_hi = float(c_lo), float(c_hi)

    def _engine(self) -> Engine:
        return _engine_for(self.reference_frame, self.c_lo, self.c_hi)

    def __call__(self, a, b):
        pts, valid, params = self._engine().catenary_points(a, b, self.length, self.n_points)
        if not valid[0]:
            return (None, None, None, None)
        return (float(params[0, 0]), float(params[0, 1]), float(params[0, 2]), pts[0])

    def batch(self, A, B):
        """(pts (n,M,3), valid (n,), params (n,3)) for n pairs in one launch."""
        return self._engine().catenary_points(A, B, self.length, self.n_points)


def compute_catenary_3D(p0, p1, rope_length, num_points):
    """models/catenary_3d.py:5-39, same name and arguments: ``(num_points, 3)`` points from ``p0`` to ``p1`` -- the hanging
    curve found by the reference's fixed-point iteration on the catenary parameter, or the straight ``np.linspace`` when the
    rope is not longer than the distance.  ``p0`` / ``p1`` of shape ``(n, 3)`` give ``(n, num_points, 3)`` in one launch."""
    a = np.asarray(p0, dtype=np.float64)
    pts, _ = default_engine().compute_catenary_3d(a, p1, rope_length, num_points)
    return pts[0] if a.ndim == 1 else pts


class Catenary3D:
    """The reference's own in-repo catenary generator (``compute_catenary_3D``, models/catenary_3d.py:5-39) behind the
    ``catenary_fn`` interface of ``transform_catenary`` (main_fun.py:63-69): ``Catenary3D(length, num_points)(a, b)`` returns a
    4-tuple whose ``[3]`` is the ``(num_points, 3)`` curve (``[0]`` = the catenary parameter, NaN when taut).  Unlike
    ``Catenary`` (a stand-in for the absent pympc class) every number it produces is pinned to reference code."""

    def __init__(self, length: float = 3.0, num_points: int = 100):
        if num_points < 2:
            raise ValueError("num_points must be >= 2")
        self.length = float(length)
        self.num_points = int(num_points)

    def __call__(self, a, b):
        pts, par = default_engine().compute_catenary_3d(a, b, self.length, self.num_points)
        return (float(par[0]), None, None, pts[0])

    def batch(self, A, B):
        """(pts (n, num_points, 3), a (n,)) for n pairs in one launch."""
        return default_engine().compute_catenary_3d(A, B, self.length, self.num_points)


def transform_catenary(point_A, point_B, catenary_fn: Callable, theta_rad, gamma_rad):
    """Returns (original, theta_rotated, theta_aligned, final) like main_fun.py:111.

    With a ``rovmpc.Catenary`` as ``catenary_fn`` the whole function is one kernel launch;
    with any other callable the two catenaries come from the callable and the per-point
    rotations (main_fun.py:96-109) run batched on the GPU."""
    A = np.asarray(point_A, dtype=np.float64); B = np.asarray(point_B, dtype=np.float64)
    if isinstance(catenary_fn, Catenary):
        out, npts, _ = catenary_fn._engine().transform_catenary(A, B, theta_rad, gamma_rad, catenary_fn.length,
                                                                catenary_fn.n_points)
        n0, n1 = int(npts[0, 0]), int(npts[0, 1])
        return out[0, 0, :n0].copy(), out[1, 0, :n1].copy(), out[2, 0, :n1].copy(), out[3, 0, :n1].copy()

    def compute_catenary(start, end):
        output = catenary_fn(start, end)
        return np.asarray(output[3], float) if output[3] is not None else np.array([start, end])

    eng = default_engine()
    original = compute_catenary(A, B)
    conn = B - A
    th_axis, ga_axis = rotation_axes(conn)
    rotated_B = A + eng.rodrigues(conn[None, :], th_axis, theta_rad)[0]
    theta_rotated = compute_catenary(A, rotated_B)
    aligned = A + eng.rodrigues(theta_rotated - A, th_axis, -theta_rad)
    final = A + eng.rodrigues(aligned - A, ga_axis, gamma_rad)
    return original, theta_rotated, aligned, final


def transform_catenary_batch(A, B, theta, gamma, catenary: Optional[Catenary] = None):
    """n cases in one launch: (out (4,n,M,3), npts (n,2), z_low (n,))."""
    cat = catenary or Catenary()
    return cat._engine().transform_catenary(A, B, theta, gamma, cat.length, cat.n_points)


def rotation_axes(connection_vector) -> Tuple[np.ndarray, np.ndarray]:
    """(theta_axis, gamma_axis) of main_fun.py:75-89,102-103 (tiny host helper)."""
    c = np.asarray(connection_vector, float)
    xy = np.array([c[0], c[1], 0.0])
    n = np.linalg.norm(xy)
    xy = np.array([1.0, 0.0, 0.0]) if n < 1e-9 else xy / n
    th = np.cross(xy, np.array([0.0, 0.0, 1.0]))
    nt = np.linalg.norm(th)
    th = np.array([0.0, 1.0, 0.0]) if nt < 1e-9 else th / nt
    return th, c / np.linalg.norm(c)


def lowest_point(points, reference_frame: str = "ENU"):
    """fully_augmented_catenary.py:21-22: (index, point) of minimum z (maximum for NED)."""
    p = np.asarray(points, float)
    idx = int(np.argmin(p[:, 2]) if reference_frame == "ENU" else np.argmax(p[:, 2]))
    return idx, p[idx]


def velocity_transform(R, v_world):
    """rob_cor_speed = R @ rob_speed per row (velocity_transform_batch.py:100-101)."""
    v = np.asarray(v_world, float)
    return default_engine().velocity_transform(R, v.reshape(-1, 3)).reshape(v.shape)


def kabsch_velocity_transform(original_points, corrected_points, rob_speed, batch_gates: bool = True):
    """``rob_cor_speed`` for T frames: Kabsch rotation of the cable markers (T, M, 3) original ->
    corrected, applied to ``rob_speed`` (T, 3).  ``batch_gates=True`` follows
    velocity_transform_batch.py:75-101 (NaN rows for non-finite markers, fewer than 3 markers or
    |P - Q| < 1e-6), ``False`` follows velocity_transform.py:60-80.  Returns (v (T,3), R (T,3,3))."""
    return default_engine().kabsch_velocity_transform(original_points, corrected_points, rob_speed, batch_gates)


def compute_rotation_kabsch(P, Q):
    """velocity_transform_batch.py:8-19 for one frame: (N,3) original and corrected points -> R (3,3)."""
    P = np.asarray(P, float); Q = np.asarray(Q, float)
    _, R = default_engine().kabsch_velocity_transform(P[None], Q[None], np.zeros((1, 3)), batch_gates=False)
    return R[0]
