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
    """The same candidate law drawn on the GPU (torch generator): the candidate tensor never exists on the host, so a
    step is one small H2D (state), two elementwise kernels, the rollout kernel and one small D2H (the record)."""

    def __init__(self, K: int, N: int, device: int = 0, dtype: str = "f64", mean=None, std=None, seed: int = 20250523):
        import torch
        self.torch = torch
        m = default_model()
        dev = torch.device("cuda", device)
        tdt = torch.float64 if dtype == "f64" else torch.float32
        self.mean = torch.tensor(np.asarray(mean if mean is not None else m.mean[3:6], float), device=dev, dtype=tdt)
        self.std = torch.tensor(np.asarray(std if std is not None else m.scale[3:6], float), device=dev, dtype=tdt)
        self.gen = torch.Generator(device=dev); self.gen.manual_seed(seed)
        self.U = torch.empty((K, N, 3), device=dev, dtype=tdt)

    def sample(self, warm_start=None):
        """Fills and returns the (K, N, 3) device tensor; ``warm_start`` (N, 3) device tensor pins candidate 0."""
        self.U.normal_(generator=self.gen).mul_(self.std).add_(self.mean)
        if warm_start is not None:
            self.U[0].copy_(warm_start)
        return self.U


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
            # candidates are drawn, rolled out and reduced on the GPU; the host sees the state and the record
            import torch
            smp = sampler if isinstance(sampler, DeviceGaussianSampler) else DeviceGaussianSampler(
                self.cfg.K, self.cfg.N, self.cfg.device, self.cfg.dtype)
            dev = torch.device("cuda", self.cfg.device)
            self._dev = {"torch": torch, "sampler": smp, "state": torch.empty(16, dtype=torch.float64, device=dev),
                         "result": torch.empty(self.engine.result_len, dtype=torch.float64, device=dev), "best": None}

    def _step_device_sampled(self, state) -> np.ndarray:
        d = self._dev; torch = d["torch"]
        ws = None
        if self.warm_start and d["best"] is not None:
            ws = torch.cat([d["best"][1:], d["best"][-1:]])                        # shifted previous optimum
        U = d["sampler"].sample(ws)
        d["state"].copy_(torch.from_numpy(state_array(state)), non_blocking=True)
        self.engine.step_device(d["state"].data_ptr(), U.data_ptr(), d["result"].data_ptr(), torch.cuda.current_stream().cuda_stream)
        rec = d["result"].cpu().numpy()                                            # [J*, k*, u(3), (theta, gamma)_0..N]
        k = int(rec[1])
        d["best"] = U[k].clone()
        self.last = StepResult(rec[2:5].copy(), rec[5:].reshape(self.cfg.N + 1, 2).copy(), float(rec[0]), k)
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
