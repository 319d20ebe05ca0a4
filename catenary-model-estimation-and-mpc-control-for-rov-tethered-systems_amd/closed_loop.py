"""Closed-loop driver: an MPC step per sample of a synthetic ROV trajectory (BASELINE config 5).

Plant model (build-defined; the reference has none): the anchor P0 follows ROV 2 and the
cable attach point P1 follows ROV 1 of the generated trajectory (Rov_traj_gen.py cases), the
measured velocity is their finite difference, and (theta, gamma) advance by the first step
of the chosen candidate's predicted trajectory.  Everything per step stays on the device:
candidates are pre-sampled in HBM, the state update is a tiny torch op.
"""
from __future__ import annotations

import time as _time
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .engine import Engine
from .trajgen import generate_rov_trajectories


@dataclass
class ClosedLoopReport:
    steps: int
    wall_s: float
    sim_s: float
    real_time_factor: float
    rollouts_per_s: float
    u: np.ndarray            # (steps, 3)
    theta_gamma: np.ndarray  # (steps + 1, 2)
    cost: np.ndarray         # (steps,)


def run_closed_loop(engine: Engine, exp_case: int = 12, n_steps: int = 1000, n_pools: int = 8,
                    seed: int = 0, sharded=None) -> ClosedLoopReport:
    import torch
    cfg = engine.cfg
    dev = torch.device("cuda", cfg.device)
    tdt = torch.float64 if cfg.dtype == "f64" else torch.float32
    total_time = n_steps * cfg.dt
    _, t0, t1 = generate_rov_trajectories(exp_case, n_steps + 1, total_time, seed=seed)
    P1 = t0[0:3].T.copy(); P0 = t1[0:3].T.copy()
    P1[:, 2] += 0.3                                            # keep the cable off the degenerate flat case
    V = np.gradient(P1, cfg.dt, axis=0) / cfg.v_scale          # m/s -> rob_cor_speed units
    A = np.gradient(V, cfg.dt, axis=0)
    mean = engine.model.mean; scale = engine.model.scale
    g = torch.Generator(device=dev); g.manual_seed(seed)
    pools = [(torch.tensor(mean[3:6], device=dev) + torch.tensor(scale[3:6], device=dev)
              * torch.randn((cfg.K, cfg.N, 3), generator=g, device=dev, dtype=torch.float64)).to(tdt).contiguous()
             for _ in range(n_pools)]
    exo = torch.tensor(np.hstack([P0, P1, V, A]), device=dev)  # (steps+1, 12)
    state = torch.empty(16, dtype=torch.float64, device=dev)
    state[12] = state[14] = float(mean[14]); state[13] = state[15] = float(mean[15])
    R = engine.result_len
    results = torch.empty((n_steps, R), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream()
    torch.cuda.synchronize()
    t_start = _time.perf_counter()
    for i in range(n_steps):
        state[0:12] = exo[i]
        if sharded is not None:
            rec = sharded.step_device(state, pools[i % n_pools])
            sharded.synchronize()
            results[i] = rec
        else:
            engine.step_device(state.data_ptr(), pools[i % n_pools].data_ptr(), results[i].data_ptr(), stream.cuda_stream)
            rec = results[i]
        # plant: theta/gamma follow the first predicted step of the chosen candidate
        state[14] = state[12]; state[15] = state[13]
        state[12] = rec[7]; state[13] = rec[8]
    torch.cuda.synchronize()
    wall = _time.perf_counter() - t_start
    res = results.cpu().numpy()
    tg = np.vstack([res[0, 5:7], res[:, 7:9]])
    sim = n_steps * cfg.dt
    return ClosedLoopReport(n_steps, wall, sim, sim / wall, n_steps * cfg.K * cfg.N / wall, res[:, 2:5], tg, res[:, 0])
