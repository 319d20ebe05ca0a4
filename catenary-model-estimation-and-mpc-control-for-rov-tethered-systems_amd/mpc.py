"""``MPC``: the shooting controller the reference's call surface implies (``step(state) -> u``).

The reference ships no solver (its ``pympc`` submodule is empty); this class supplies one
whose per-step work is the fused HIP rollout kernel: sample / take K candidate control
sequences, roll all of them over the horizon, return the first control of the cheapest.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .engine import Engine, MPCConfig, MPCState, StepResult, state_array
from .model import DynamicsModel, default_model


class GaussianSampler:
    """U[k, n, :] = mean + std * N(0, 1), i.i.d.; candidate 0 may be pinned to a warm start."""

    def __init__(self, mean=None, std=None, seed: int = 20250523):
        m = default_model()
        self.mean = np.asarray(mean if mean is not None else m.mean[3:6], float)      # scaler x3..x5
        self.std = np.asarray(std if std is not None else m.scale[3:6], float)
        self.rng = np.random.default_rng(seed)

    def sample(self, K: int, N: int, dtype=np.float64, warm_start: Optional[np.ndarray] = None) -> np.ndarray:
        U = (self.mean + self.std * self.rng.standard_normal((K, N, 3))).astype(dtype, copy=False)
        if warm_start is not None:
            U[0] = warm_start
        return U


class DeviceGaussianSampler:
    """The same candidate law drawn on the GPU by the library itself (``rovmpc_mpc_step_sampled``): standard normals from
    Philox4x32-10 keyed by (seed, step counter) through Box-Muller, so a step is a pure function of (state, seed, step) and
    reproducible on the host (the test oracle restates the law).  The candidate tensor never exists on the host."""

    def __init__(self, mean=None, std=None, seed: int = 20250523):
        m = default_model()
        self.mean = np.ascontiguousarray(mean if mean is not None else m.mean[3:6], dtype=np.float64)
        self.std = np.ascontiguousarray(std if std is not None else m.scale[3:6], dtype=np.float64)
        self.seed = int(seed)
        self.step = 0


class MPC:
    """``mpc = MPC(N=20, K=4096); u = mpc.step(state)``.

    ``state``: :class:`MPCState`, a dict of its fields, or 16 floats
    (P0, P1, V1, A1, theta, gamma, theta_prev, gamma_prev).
    After ``step`` the full result (u, predicted (theta, gamma) trajectory, cost, index) is in
    ``mpc.last``.  Pass ``U`` (K, N, 3) to evaluate given candidates instead of sampling.
    """

    def __init__(self, cfg: Optional[MPCConfig] = None, model: Optional[DynamicsModel] = None,
                 sampler: Optional[GaussianSampler] = None, warm_start: bool = True, device_sampling: bool = False,
                 **overrides):
        self.engine = Engine(cfg, model, **overrides)
        self.cfg = self.engine.cfg
        self.sampler = sampler or GaussianSampler()
        self.warm_start = warm_start
        self.last: Optional[StepResult] = None
        self._best_seq: Optional[np.ndarray] = None
        self._dev = None
        if device_sampling:
            # candidates are drawn, rolled out and reduced on the GPU by one library call; the host sees the state and the record
            self._dev = sampler if isinstance(sampler, DeviceGaussianSampler) else DeviceGaussianSampler()

    def _step_device_sampled(self, state) -> np.ndarray:
        d = self._dev
        rec = self.engine.mpc_step_sampled(state, d.seed, d.step, d.mean, d.std, self.warm_start)
        d.step += 1
        N = self.cfg.N
        self.last = StepResult(rec[2:5].copy(), rec[5:].reshape(N + 1, 2).copy(), float(rec[0]), int(rec[1]))
        return self.last.u

    def step(self, state, U: Optional[np.ndarray] = None) -> np.ndarray:
        if U is None and self._dev is not None:
            return self._step_device_sampled(state)
        if U is None:
            ws = None
            if self.warm_start and self._best_seq is not None:
                ws = np.vstack([self._best_seq[1:], self._best_seq[-1:]])          # shifted previous optimum
            U = self.sampler.sample(self.cfg.K, self.cfg.N, self.cfg.np_dtype, ws)
        res = self.engine.step(state, U)
        self.last = res
        self._best_seq = np.asarray(U[res.index], dtype=np.float64).copy()
        return res.u

    def rollout_costs(self, state, U, return_traj: bool = False):
        return self.engine.rollout_costs(state, U, return_traj)

    def close(self):
        self.engine.close()


def synthetic_problem(K: int, N: int, seed: int = 20250523, dtype=np.float64):
    """The synthetic MPC step of SURVEY section 8(d) / BASELINE.md section 3: state at the scaler means,
    candidates drawn from the scaler statistics of x3..x5.  Returns (state(16,), U(K,N,3))."""
    m = default_model()
    rng = np.random.default_rng(seed)
    P1 = m.mean[0:3] + 0.05 * rng.standard_normal(3)
    st = MPCState(P0=(0.0, 0.0, 0.0), P1=P1, V1=m.mean[3:6], A1=(0.0, 0.0, 0.0),
                  theta=-0.0342, gamma=-0.0522, theta_prev=-0.0342, gamma_prev=-0.0522)
    U = m.mean[3:6] + m.scale[3:6] * rng.standard_normal((K, N, 3))
    return st.as_array(), U.astype(dtype, copy=False)
