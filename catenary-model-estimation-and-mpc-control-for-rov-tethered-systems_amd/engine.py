"""``Engine``: one librovmpc handle (one GPU, one stream, one workspace) with NumPy / device
pointer entry points.  Everything numeric happens in the HIP library."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field, asdict
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import Config, State, RovmpcError, check, load_library
from .model import DynamicsModel, default_model


@dataclass
class MPCConfig:
    """Mirror of ``rovmpc_config`` (include/rovmpc.h).  Defaults = SURVEY section 8(d) synthetic
    set-up, except dt (1/60 s: the learned gamma equation is a finite difference at the
    data rate) and non-zero w_T / w_floor so every cost term is live."""
    N: int = 20
    K: int = 4096
    dtype: str = "f64"
    device: int = 0
    n_shape_pts: int = 16
    vt_mode: int = _lib.VT_COMPOSE
    prev_mode: int = _lib.PREV_INTERP
    integrator: int = _lib.RK4
    frame: str = "ENU"
    force_interpreter: bool = False
    candidates_per_block: int = 0
    threads_per_block: int = 0
    debug_flags: int = 0
    jit: bool = True             # specialise non-default models with hiprtc at set_model
    no_builtin: bool = False     # never substitute the compiled-in kernel for the reference's chosen rows
    feature_map: int = _lib.FEATURES_GEN1   # FEATURES_GEN2: 17 unscaled slots of simulate_rk4_theta_gamma.py:12-42
    dt: float = 1.0 / 60.0
    v_scale: float = 1e-3
    L: float = 3.0
    cable_wet_weight: float = 1.521
    c_lo: float = 1e-6
    c_hi: float = 10.0
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

    def to_c(self) -> Config:
        if self.dtype not in ("f64", "f32"):
            raise ValueError("dtype must be 'f64' or 'f32'")
        if self.frame not in ("ENU", "NED"):
            raise ValueError("frame must be 'ENU' or 'NED'")
        c = Config()
        c.struct_size = C.sizeof(Config)
        c.device = self.device
        c.dtype = _lib.F64 if self.dtype == "f64" else _lib.F32
        c.N, c.K, c.n_shape_pts = self.N, self.K, self.n_shape_pts
        c.vt_mode, c.prev_mode, c.integrator = self.vt_mode, self.prev_mode, self.integrator
        c.frame = _lib.ENU if self.frame == "ENU" else _lib.NED
        c.force_interpreter = int(self.force_interpreter)
        c.candidates_per_block = self.candidates_per_block
        c.threads_per_block = self.threads_per_block
        c.debug_flags = self.debug_flags
        c.jit_off = 0 if self.jit else 1
        c.no_builtin = int(self.no_builtin)
        c.feature_map = self.feature_map
        for k in ("dt", "v_scale", "L", "cable_wet_weight", "c_lo", "c_hi", "w_theta", "w_gamma", "w_u", "w_T",
                  "w_taut", "rho_taut", "w_floor", "z_floor", "theta_ref", "gamma_ref"):
            setattr(c, k, float(getattr(self, k)))
        c.U_ref = (C.c_double * 3)(*[float(v) for v in self.U_ref])
        return c

    @property
    def np_dtype(self):
        return np.float64 if self.dtype == "f64" else np.float32


@dataclass
class MPCState:
    """Feature-slot values at horizon node 0 (``rovmpc_state``): P0 rod_end [m], P1 ROV cable
    attach point [m], V1 rob_cor_speed, A1 its time derivative, theta, gamma and their
    previous samples (simply.py:15-41)."""
    P0: Sequence[float] = (0.0, 0.0, 0.0)
    P1: Sequence[float] = (0.0, 0.0, 0.0)
    V1: Sequence[float] = (0.0, 0.0, 0.0)
    A1: Sequence[float] = (0.0, 0.0, 0.0)
    theta: float = 0.0
    gamma: float = 0.0
    theta_prev: Optional[float] = None
    gamma_prev: Optional[float] = None

    def as_array(self) -> np.ndarray:
        tp = self.theta if self.theta_prev is None else self.theta_prev
        gp = self.gamma if self.gamma_prev is None else self.gamma_prev
        a = np.concatenate([np.asarray(self.P0, float), np.asarray(self.P1, float), np.asarray(self.V1, float),
                            np.asarray(self.A1, float), [self.theta, self.gamma, tp, gp]])
        if a.shape != (16,):
            raise ValueError("P0, P1, V1, A1 must have 3 components each")
        return a


def state_array(state) -> np.ndarray:
    if isinstance(state, MPCState):
        return state.as_array()
    if isinstance(state, dict):
        return MPCState(**state).as_array()
    a = np.ascontiguousarray(state, dtype=np.float64).reshape(-1)
    if a.shape != (16,):
        raise ValueError("state must be an MPCState, a dict of its fields or 16 floats")
    return a


def _c_state(a: np.ndarray) -> State:
    s = State()
    C.memmove(C.byref(s), a.ctypes.data, 16 * 8)
    return s


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class StepResult:
    u: np.ndarray          # (3,)  first control of the best candidate
    traj: np.ndarray       # (N+1, 2) predicted (theta, gamma)
    cost: float
    index: int


class Engine:
    def __init__(self, cfg: Optional[MPCConfig] = None, model: Optional[DynamicsModel] = None, **overrides):
        self.lib = load_library()
        self.cfg = cfg or MPCConfig()
        for k, v in overrides.items():
            if not hasattr(self.cfg, k):
                raise TypeError(f"unknown config field {k!r}")
            setattr(self.cfg, k, v)
        self._h = C.c_void_p()
        self.has_comm = False          # rovmpc_comm_init has been called: steps of the closed-loop driver are the sharded ones
        c = self.cfg.to_c()
        rc = self.lib.rovmpc_create(C.byref(c), C.byref(self._h))
        if rc != 0:
            msg = self.lib.rovmpc_last_error(None)
            raise RovmpcError(rc, msg.decode() if msg else "")
        self.model = None
        self.set_model(model or default_model())

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.rovmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        check(self.lib, self._h, rc)

    # -- model -----------------------------------------------------------------------------
    def set_model(self, model: DynamicsModel):
        ct = np.asarray(model.prog_theta.code, dtype=np.int32)
        cg = np.asarray(model.prog_gamma.code, dtype=np.int32)
        cs = np.asarray(model.consts if model.consts else [0.0], dtype=np.float64)
        self._check(self.lib.rovmpc_set_model(self._h, model.n_features, _ptr(model.mean), _ptr(model.scale),
                                              _ptr(ct), len(ct), _ptr(cg), len(cg), _ptr(cs), len(model.consts)))
        self.model = model
        note = self.lib.rovmpc_last_error(self._h)
        self.model_note = note.decode() if note else ""     # e.g. why hiprtc specialisation was not used

    @property
    def model_path(self) -> str:
        """'builtin' (compiled-in reference rows), 'interpreter' or 'jit' (hiprtc-specialised)."""
        return ("builtin", "interpreter", "jit")[int(self.lib.rovmpc_model_path(self._h))]

    @property
    def model_structure(self) -> set:
        """What the code generator found in the loaded rows: {"gamma_invariant"} when dgamma/dt reads only (gamma, gamma_prev)
        -- the gamma path is then integrated once per workgroup --, {"theta_stage_free"} when dtheta/dt reads no stage state."""
        b = int(self.lib.rovmpc_model_structure(self._h))
        return {n for k, n in ((1, "gamma_invariant"), (2, "theta_stage_free")) if b & k}

    def set_rotation_table(self, R):
        R = np.ascontiguousarray(R, dtype=np.float64)
        if R.shape != (self.cfg.N, 3, 3):
            raise ValueError(f"R must have shape ({self.cfg.N}, 3, 3)")
        self._check(self.lib.rovmpc_set_rotation_table(self._h, _ptr(R)))

    # -- hot path, host buffers ------------------------------------------------------------
    def _U(self, U) -> np.ndarray:
        U = np.ascontiguousarray(U, dtype=self.cfg.np_dtype)
        if U.shape != (self.cfg.K, self.cfg.N, 3):
            raise ValueError(f"U must have shape ({self.cfg.K}, {self.cfg.N}, 3), got {U.shape}")
        return U

    def step(self, state, U) -> StepResult:
        sa = state_array(state)
        U = self._U(U)
        u = np.empty(3); traj = np.empty((self.cfg.N + 1, 2))
        cost = C.c_double(); idx = C.c_int64()
        s = _c_state(sa)
        self._check(self.lib.rovmpc_step(self._h, C.byref(s), _ptr(U), _ptr(u), _ptr(traj), C.byref(cost), C.byref(idx)))
        return StepResult(u, traj, cost.value, idx.value)

    def mpc_step_sampled(self, state, seed: int, step: int, mean, std, warm_start: bool = True) -> np.ndarray:
        """One control step with the candidates drawn on the GPU (Philox4x32-10 keyed by (seed, step), Box-Muller):
        returns the record [J*, k*, u(3), (theta, gamma)_0..N].  One library call; the host sends the state, spins on the
        completion word and reads the record from mapped memory."""
        if getattr(self, "_samp", None) is None:
            self._samp = {"state": State(), "rec": np.empty(self.result_len), "mean": np.empty(3), "std": np.empty(3)}
            self._samp["prec"] = _ptr(self._samp["rec"]); self._samp["pm"] = _ptr(self._samp["mean"]); self._samp["ps"] = _ptr(self._samp["std"])
            self._samp["pstate"] = C.byref(self._samp["state"])
        sp = self._samp
        sa = state if (isinstance(state, np.ndarray) and state.dtype == np.float64 and state.shape == (16,) and state.flags.c_contiguous) else state_array(state)
        C.memmove(sp["pstate"], sa.ctypes.data, 128)
        sp["mean"][:] = mean; sp["std"][:] = std
        rc = self.lib.rovmpc_mpc_step_sampled(self._h, sp["pstate"], seed, step, sp["pm"], sp["ps"], int(warm_start), sp["prec"])
        if rc:
            self._check(rc)
        return sp["rec"]

    def sampled_candidates(self) -> np.ndarray:
        """Host copy of the candidate tensor of the last ``mpc_step_sampled`` (tests / inspection)."""
        U = np.empty((self.cfg.K, self.cfg.N, 3), dtype=self.cfg.np_dtype)
        self._check(self.lib.rovmpc_sampled_candidates(self._h, _ptr(U)))
        return U

    def sample_candidates_device(self, seed: int, step: int, mean, std, d_U: int, stream: int = 0):
        m = np.ascontiguousarray(mean, np.float64); s = np.ascontiguousarray(std, np.float64)
        self._check(self.lib.rovmpc_sample_candidates_device(self._h, seed, step, _ptr(m), _ptr(s), d_U, stream))

    def rollout_costs(self, state, U, return_traj: bool = False):
        sa = state_array(state)
        U = self._U(U)
        J = np.empty(self.cfg.K, dtype=self.cfg.np_dtype)
        traj = np.empty((self.cfg.K, self.cfg.N + 1, 2), dtype=self.cfg.np_dtype) if return_traj else None
        s = _c_state(sa)
        self._check(self.lib.rovmpc_rollout_costs(self._h, C.byref(s), _ptr(U), _ptr(J), _ptr(traj)))
        return (J, traj) if return_traj else J

    # -- hot path, device buffers (raw pointers; torch tensors' data_ptr()) ----------------
    @property
    def result_len(self) -> int:
        return int(self.lib.rovmpc_result_len(self._h))

    def step_device(self, d_state: int, d_U: int, d_result: int, stream: int = 0):
        self._check(self.lib.rovmpc_step_device(self._h, d_state, d_U, d_result, stream))

    def step_device_sharded(self, d_state: int, d_U: int, k_offset: int, rank: int, world: int, d_slots: int,
                            stream: int = 0):
        self._check(self.lib.rovmpc_step_device_sharded(self._h, d_state, d_U, k_offset, rank, world, d_slots, stream))

    def select_device(self, d_slots: int, world: int, d_result: int, stream: int = 0):
        self._check(self.lib.rovmpc_select_device(self._h, d_slots, world, d_result, stream))

    # -- native collective (RCCL called from the library) -----------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        rc = self.lib.rovmpc_comm_unique_id(buf)
        if rc != 0:
            msg = self.lib.rovmpc_last_error(None)
            raise RovmpcError(rc, msg.decode() if msg else "")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be 128 bytes")
        self._check(self.lib.rovmpc_comm_init(self._h, C.create_string_buffer(unique_id, 128), rank, world))
        self.has_comm = True

    def step_device_allreduce(self, d_state: int, d_U: int, k_offset: int, d_result: int, stream: int = 0):
        self._check(self.lib.rovmpc_step_device_allreduce(self._h, d_state, d_U, k_offset, d_result, stream))

    def comm_join(self, stream: int = 0):
        self._check(self.lib.rovmpc_comm_join(self._h, stream))

    def comm_sync(self, stream: int = 0):
        """Join + synchronise ``stream``; raises ``RovmpcError`` if a GPU-side hand-off of any enqueued step gave up."""
        self._check(self.lib.rovmpc_comm_sync(self._h, stream))

    def comm_abort(self):
        """ncclCommAbort on the handle's communicators (callable from another thread while a synchronise is stuck)."""
        self._check(self.lib.rovmpc_comm_abort(self._h))

    def comm_placement(self) -> str:
        """One-line report of the collective-stream placement probe (empty before the first sharded step)."""
        v = self.lib.rovmpc_comm_placement(self._h)
        return v.decode() if v else ""

    def set_option(self, name: str, value: float):
        self._check(self.lib.rovmpc_set_option(self._h, name.encode(), float(value)))

    def device_status(self):
        self._check(self.lib.rovmpc_device_status(self._h))

    def step_batch_device(self, B: int, d_states: int, d_U: int, d_results: int, stream: int = 0):
        """B independent problems in one launch: d_states[B][16], d_U[B][K][N][3], d_results[B][result_len]."""
        self._check(self.lib.rovmpc_step_batch_device(self._h, B, d_states, d_U, d_results, stream))

    def batch_costs_ptr(self) -> int:
        p = C.c_void_p()
        self._check(self.lib.rovmpc_batch_costs_device(self._h, C.byref(p)))
        return p.value

    def comm_destroy(self):
        self.has_comm = False
        self._check(self.lib.rovmpc_comm_destroy(self._h))

    def closed_loop_device(self, d_exo: int, T: int, d_state: int, d_pools: int, n_pools: int, d_results: int,
                           k_offset: int = 0, feedback: bool = False, stream: int = 0, mode: str = ""):
        """mode: "per_step" (one launch per step, one stream; also the sharded loop) or "pipelined" (one launch per step,
        two streams, GPU-side state hand-off; single GPU)."""
        mode = mode or "per_step"
        if mode == "pipelined":
            if k_offset:
                raise ValueError("the pipelined loop is single-GPU (k_offset must be 0)")
            self._check(self.lib.rovmpc_closed_loop_pipelined_device(self._h, d_exo, T, d_state, d_pools, n_pools, int(feedback),
                                                                     d_results, stream))
            return
        if mode != "per_step":
            raise ValueError(f"unknown closed-loop mode {mode!r}")
        self._check(self.lib.rovmpc_closed_loop_device(self._h, d_exo, T, d_state, d_pools, n_pools, k_offset,
                                                       int(feedback), d_results, stream))

    def timing_enable(self, max_launches: int):
        self._check(self.lib.rovmpc_timing_enable(self._h, max_launches))

    def timing_read(self):
        avg = C.c_double(); mn = C.c_double(); n = C.c_int32()
        self._check(self.lib.rovmpc_timing_read(self._h, C.byref(avg), C.byref(mn), C.byref(n)))
        return avg.value, mn.value, n.value

    # -- batched helper mirrors ------------------------------------------------------------
    def predict(self, Xs, which: int) -> np.ndarray:
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        if Xs.ndim != 2 or Xs.shape[1] != self.model.n_features:
            raise ValueError(f"X must be (n, {self.model.n_features})")
        out = np.empty(Xs.shape[0])
        self._check(self.lib.rovmpc_predict(self._h, _ptr(Xs), Xs.shape[0], which, _ptr(out)))
        return out

    def eval_expression(self, code, consts, X) -> np.ndarray:
        """A compiled expression (``expr.compile_expression``) on every row of X (n, F)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("X must be (n, F)")
        c = np.asarray(code, dtype=np.int32); k = np.asarray(consts if len(consts) else [0.0], dtype=np.float64)
        out = np.empty(X.shape[0])
        self._check(self.lib.rovmpc_eval_expression(self._h, _ptr(c), len(c), _ptr(k), len(consts), _ptr(X), X.shape[1], X.shape[0], _ptr(out)))
        return out

    def lagrangian_rollout(self, code_theta, code_gamma, consts, time, y0):
        """evaluate_lagrangian_on_test.py:59-68 for B initial states y0 (B, 4) = (theta, gamma, vtheta, vgamma): four (B, T) arrays."""
        time = np.ascontiguousarray(time, dtype=np.float64).reshape(-1)
        y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(-1, 4)
        ct = np.asarray(code_theta, dtype=np.int32); cg = np.asarray(code_gamma, dtype=np.int32)
        k = np.asarray(consts if len(consts) else [0.0], dtype=np.float64)
        out = np.empty((4, y0.shape[0], time.shape[0]))
        self._check(self.lib.rovmpc_lagrangian_rollout(self._h, _ptr(ct), len(ct), _ptr(cg), len(cg), _ptr(k), len(consts), _ptr(time),
                                                       time.shape[0], _ptr(y0), y0.shape[0], _ptr(out)))
        return out[0], out[1], out[2], out[3]

    def replay(self, Xs, time, theta0: float, gamma0: float, integrator: int = _lib.RK4):
        Xs = np.ascontiguousarray(Xs, dtype=np.float64); time = np.ascontiguousarray(time, dtype=np.float64)
        if Xs.ndim != 2 or Xs.shape[1] != self.model.n_features or time.shape != (Xs.shape[0],):
            raise ValueError("x_input must be (T, n_features) and time (T,)")
        th = np.empty(len(time)); ga = np.empty(len(time))
        self._check(self.lib.rovmpc_replay(self._h, _ptr(Xs), _ptr(time), len(time), float(theta0), float(gamma0),
                                           integrator, _ptr(th), _ptr(ga)))
        return th, ga

    def solve_catenary(self, l, delta_H, L: float, with_tension: bool = False):
        l, dH = np.broadcast_arrays(np.asarray(l, np.float64), np.asarray(delta_H, np.float64))
        shape = l.shape
        l = np.ascontiguousarray(l).reshape(-1); dH = np.ascontiguousarray(dH).reshape(-1)
        Cc = np.empty(l.size); T = np.empty(l.size) if with_tension else None
        self._check(self.lib.rovmpc_solve_catenary(self._h, _ptr(l), _ptr(dH), float(L), l.size, _ptr(Cc), _ptr(T)))
        return (Cc.reshape(shape), T.reshape(shape)) if with_tension else Cc.reshape(shape)

    def rodrigues(self, v, axis, angle) -> np.ndarray:
        v = np.ascontiguousarray(v, np.float64).reshape(-1, 3)
        axis = np.ascontiguousarray(np.broadcast_to(np.asarray(axis, np.float64), v.shape))
        angle = np.ascontiguousarray(np.broadcast_to(np.asarray(angle, np.float64), (v.shape[0],)))
        out = np.empty_like(v)
        self._check(self.lib.rovmpc_rodrigues(self._h, _ptr(v), _ptr(axis), _ptr(angle), v.shape[0], _ptr(out)))
        return out

    def catenary_points(self, A, B, L: float, M: int):
        A = np.ascontiguousarray(A, np.float64).reshape(-1, 3); B = np.ascontiguousarray(B, np.float64).reshape(-1, 3)
        n = A.shape[0]
        pts = np.empty((n, M, 3)); valid = np.empty(n, np.int32); params = np.empty((n, 3))
        self._check(self.lib.rovmpc_catenary_points(self._h, _ptr(A), _ptr(B), float(L), n, M, _ptr(pts), _ptr(valid),
                                                    _ptr(params)))
        return pts, valid.astype(bool), params

    def compute_catenary_3d(self, p0, p1, rope_length: float, num_points: int):
        """models/catenary_3d.py:5-39 for n pairs: (pts (n, num_points, 3), a (n,)); a = NaN where the rope is taut."""
        A = np.ascontiguousarray(p0, np.float64).reshape(-1, 3); B = np.ascontiguousarray(p1, np.float64).reshape(-1, 3)
        if A.shape != B.shape:
            raise ValueError("p0 and p1 must have the same shape")
        if num_points < 2:
            raise ValueError("num_points must be >= 2")
        n = A.shape[0]
        pts = np.empty((n, num_points, 3)); a = np.empty(n)
        self._check(self.lib.rovmpc_compute_catenary_3d(self._h, _ptr(A), _ptr(B), float(rope_length), n, int(num_points),
                                                        _ptr(pts), _ptr(a)))
        return pts, a

    def transform_catenary(self, A, B, theta, gamma, L: float, M: int):
        A = np.ascontiguousarray(A, np.float64).reshape(-1, 3); B = np.ascontiguousarray(B, np.float64).reshape(-1, 3)
        n = A.shape[0]
        th = np.ascontiguousarray(np.broadcast_to(np.asarray(theta, np.float64), (n,)))
        ga = np.ascontiguousarray(np.broadcast_to(np.asarray(gamma, np.float64), (n,)))
        out = np.empty((4, n, M, 3)); npts = np.empty((n, 2), np.int32); z = np.empty(n)
        self._check(self.lib.rovmpc_transform_catenary(self._h, _ptr(A), _ptr(B), _ptr(th), _ptr(ga), float(L), n, M,
                                                       _ptr(out), _ptr(npts), _ptr(z)))
        return out, npts, z

    def velocity_transform(self, R, v) -> np.ndarray:
        R = np.ascontiguousarray(R, np.float64).reshape(-1, 3, 3); v = np.ascontiguousarray(v, np.float64).reshape(-1, 3)
        if R.shape[0] != v.shape[0]:
            raise ValueError("R and v must have the same number of rows")
        out = np.empty_like(v)
        self._check(self.lib.rovmpc_velocity_transform(self._h, _ptr(R), _ptr(v), v.shape[0], _ptr(out)))
        return out


    def extract_features(self, P0, P1, V1, time, theta, gamma, with_prev: bool = True) -> np.ndarray:
        c = lambda x, shape: np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.float64), shape))
        time = np.ascontiguousarray(time, np.float64).reshape(-1)
        T = time.shape[0]
        P0, P1, V1 = c(P0, (T, 3)), c(P1, (T, 3)), c(V1, (T, 3))
        theta, gamma = c(np.asarray(theta, np.float64).reshape(-1), (T,)), c(np.asarray(gamma, np.float64).reshape(-1), (T,))
        out = np.empty((T, 18 if with_prev else 16))
        self._check(self.lib.rovmpc_extract_features(self._h, _ptr(P0), _ptr(P1), _ptr(V1), _ptr(time), _ptr(theta),
                                                     _ptr(gamma), T, int(with_prev), _ptr(out)))
        return out

    def features_dd(self, P0_mm, P1_mm, V_mm, time, theta, gamma, window: int = 11, polyorder: int = 3):
        """main_fun.py:811-871 on arrays as logged (mm, mm/s): returns (features (T,14), targets (T,2))."""
        c = lambda x, shape: np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.float64), shape))
        time = np.ascontiguousarray(time, np.float64).reshape(-1)
        T = time.shape[0]
        if T < window:
            raise ValueError("If mode is 'interp', window_length must be less than or equal to the size of x.")   # scipy's message
        P0, P1, V = c(P0_mm, (T, 3)), c(P1_mm, (T, 3)), c(V_mm, (T, 3))
        theta, gamma = c(np.asarray(theta, np.float64).reshape(-1), (T,)), c(np.asarray(gamma, np.float64).reshape(-1), (T,))
        F = np.empty((T, 14)); Y = np.empty((T, 2))
        self._check(self.lib.rovmpc_features_dd(self._h, _ptr(P0), _ptr(P1), _ptr(V), _ptr(time), _ptr(theta), _ptr(gamma),
                                                T, window, polyorder, _ptr(F), _ptr(Y)))
        return F, Y

    def gaussian_filter1d(self, x, sigma: float, truncate: float = 4.0) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float64).reshape(-1)
        out = np.empty_like(x)
        self._check(self.lib.rovmpc_gaussian_filter1d(self._h, _ptr(x), x.shape[0], float(sigma), float(truncate), _ptr(out)))
        return out

    def kabsch_velocity_transform(self, P, Q, v, batch_gates: bool = True):
        P = np.ascontiguousarray(P, np.float64); Q = np.ascontiguousarray(Q, np.float64)
        v = np.ascontiguousarray(v, np.float64).reshape(-1, 3)
        if P.ndim != 3 or P.shape[2] != 3 or Q.shape != P.shape or v.shape[0] != P.shape[0]:
            raise ValueError("P, Q must be (T, M, 3) and v (T, 3)")
        T, M = P.shape[0], P.shape[1]
        out = np.empty((T, 3)); R = np.empty((T, 3, 3))
        self._check(self.lib.rovmpc_kabsch_velocity_transform(self._h, _ptr(P), _ptr(Q), _ptr(v), T, M, int(batch_gates),
                                                              _ptr(out), _ptr(R)))
        return out, R


_default_engine: Optional[Engine] = None


def default_engine() -> Engine:
    """Small shared handle for the stateless helper mirrors (geometry, predict, replay)."""
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(MPCConfig(N=1, K=1))
    return _default_engine
