"""Closed-loop driver: an MPC step per sample of a synthetic ROV trajectory (BASELINE config 5).

Plant model (build-defined; the reference has none): the anchor P0 follows ROV 2 and the cable
attach point P1 follows ROV 1 of the generated trajectory (Rov_traj_gen.py cases), the measured
velocity / acceleration are their finite differences, and (theta, gamma) advance to the first
predicted node of the chosen candidate.  Everything per step stays on the device: candidate
batches are pre-sampled in HBM, the plant update is a one-workgroup kernel, and the whole loop is
enqueued by ONE library call (``rovmpc_closed_loop_device``) with no host synchronisation.
"""
from __future__ import annotations

import time as _time
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
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
    index: np.ndarray = None  # (steps,) chosen candidate (global index when sharded)


def closed_loop_inputs(engine: Engine, exp_case: int, n_steps: int, seed: int = 0):
    """(rows (n_steps, 16) in rovmpc_state order, initial state (16,)).  P0/P1/V1/A1 come from the
    generated two-ROV trajectory; the measured (theta, gamma) are a smooth bounded synthetic signal
    around the scaler means (the reference's recorded angles are not in the snapshot)."""
    cfg = engine.cfg
    total_time = n_steps * cfg.dt
    t, t0, t1 = generate_rov_trajectories(exp_case, n_steps + 1, total_time, seed=seed)
    P1 = t0[0:3].T.copy(); P0 = t1[0:3].T.copy()
    P1[:, 2] += 0.3                                            # keep the cable off the degenerate flat case
    V = np.gradient(P1, cfg.dt, axis=0) / cfg.v_scale          # m/s -> rob_cor_speed units
    A = np.gradient(V, cfg.dt, axis=0)
    mean, scale = engine.model.mean, engine.model.scale
    it, ig = angle_slots(cfg.feature_map)
    th = mean[it] + scale[it] * np.sin(2 * np.pi * t / 7.0)
    ga = mean[ig] + scale[ig] * np.cos(2 * np.pi * t / 11.0)
    if cfg.feature_map == _lib.FEATURES_GEN3:          # second-order map: slots 14/15 of the state carry the rates
        thp = np.gradient(th, cfg.dt); gap = np.gradient(ga, cfg.dt)
    else:
        thp = np.roll(th, 1); gap = np.roll(ga, 1); thp[0] = th[0]; gap[0] = ga[0]
    rows = np.ascontiguousarray(np.hstack([P0, P1, V, A, th[:, None], ga[:, None], thp[:, None], gap[:, None]])[:n_steps])
    return rows, rows[0].copy()


def angle_slots(feature_map: int):
    """(theta slot, gamma slot) of the scaler for a feature map."""
    return {_lib.FEATURES_GEN1: (14, 15), _lib.FEATURES_GEN2: (12, 13), _lib.FEATURES_GEN3: (0, 1)}[feature_map]


def velocity_prior(engine: Engine):
    """(mean(3), scale(3)) of the candidate controls in rob_cor_speed units, from the scaler's velocity slots."""
    m, s = engine.model.mean, engine.model.scale
    if engine.cfg.feature_map == _lib.FEATURES_GEN3:   # V_x..V_z are slots 8..10, in m/s
        return m[8:11] / engine.cfg.v_scale, s[8:11] / engine.cfg.v_scale
    if engine.cfg.feature_map == _lib.FEATURES_GEN2:   # unscaled model: the generation-1 spread
        return np.array([80.85, -20.13, -18.35]), np.array([108.49, 15.88, 63.13])
    return m[3:6], s[3:6]


def closed_loop_pools(engine: Engine, n_pools: int = 8, seed: int = 0, k_offset: int = 0):
    """The candidate batches the loop cycles through, (n_pools, K, N, 3) on the engine's device: i.i.d. normal around the
    scaler's velocity statistics, drawn on the GPU (a shard draws its own candidates: the seed moves with k_offset)."""
    import torch
    cfg = engine.cfg
    dev = torch.device("cuda", cfg.device)
    tdt = torch.float64 if cfg.dtype == "f64" else torch.float32
    vm, vs = velocity_prior(engine)
    mean = torch.tensor(np.ascontiguousarray(vm), device=dev); scale = torch.tensor(np.ascontiguousarray(vs), device=dev)
    g = torch.Generator(device=dev); g.manual_seed(seed + 1000 * k_offset)
    return (mean + scale * torch.randn((n_pools, cfg.K, cfg.N, 3), generator=g, device=dev, dtype=torch.float64)).to(tdt).contiguous()


def state_of_step(rows: np.ndarray, report: "ClosedLoopReport", i: int, feedback: bool) -> np.ndarray:
    """The 16-double state step i of a recorded loop started from (the plant rule of rovmpc_closed_loop_device): the measured
    row, and with feedback from the second step on (theta, gamma) = first predicted node of step i - 1's winner,
    (theta_prev, gamma_prev) = the (theta, gamma) step i - 1 started from."""
    st = np.array(rows[i], dtype=np.float64)
    if feedback and i > 0:
        st[12:14] = report.theta_gamma[i]
        st[14:16] = report.theta_gamma[i - 1]
    return st


def run_closed_loop(engine: Engine, exp_case: int = 12, n_steps: int = 1000, n_pools: int = 8,
                    seed: int = 0, k_offset: int = 0, feedback: bool = False, mode: str = "", pools=None) -> ClosedLoopReport:
    """Runs the loop on ``engine``'s device.  If the engine has a native communicator
    (``Engine.comm_init``) every step is the candidate-sharded one.  ``feedback=True`` replaces the
    measured (theta, gamma) by the model's own first predicted node from the second step on.
    ``pools``: the candidate batches to cycle through (device tensor (n_pools, K, N, 3)); default ``closed_loop_pools``."""
    import torch
    cfg = engine.cfg
    dev = torch.device("cuda", cfg.device)
    exo_np, state_np = closed_loop_inputs(engine, exp_case, n_steps, seed)
    if pools is None:
        pools = closed_loop_pools(engine, n_pools, seed, k_offset)
    n_pools = int(pools.shape[0])
    exo = torch.tensor(exo_np, device=dev)
    state = torch.tensor(state_np, device=dev)
    R = engine.result_len
    results = torch.empty((n_steps, R), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream()
    torch.cuda.synchronize()
    t_start = _time.perf_counter()
    if mode == "batched":
        # measured rows only: no step depends on another, so n_pools consecutive steps (states = rows i .. i + n_pools - 1,
        # candidates = the n_pools batches in order) are ONE batched launch -- the evaluation of a controller along a recorded
        # trajectory, which is what the reference's scripts do with their data; same records as the step-by-step forms
        if feedback or k_offset:
            raise ValueError("mode 'batched' replays measured rows on one GPU: feedback and sharding make the steps sequential")
        i = 0
        while i < n_steps:
            B = min(n_pools, n_steps - i) if i % n_pools == 0 else 1
            engine.step_batch_device(B, exo[i].data_ptr(), pools[i % n_pools].data_ptr(), results[i].data_ptr(), stream.cuda_stream)
            i += B
    else:
        engine.closed_loop_device(exo.data_ptr(), n_steps, state.data_ptr(), pools.data_ptr(), n_pools, results.data_ptr(),
                                  k_offset, feedback, stream.cuda_stream, mode=mode)
    if engine.has_comm:
        engine.comm_sync(stream.cuda_stream)            # raises when a GPU-side hand-off of any step gave up
    torch.cuda.synchronize()
    wall = _time.perf_counter() - t_start
    res = results.cpu().numpy()
    tg = np.vstack([res[0, 5:7], res[:, 7:9]])                  # start state, then each step's first predicted node
    sim = n_steps * cfg.dt
    return ClosedLoopReport(n_steps, wall, sim, sim / wall, n_steps * cfg.K * cfg.N / wall, res[:, 2:5], tg, res[:, 0], res[:, 1].astype(np.int64))


def run_closed_loop_sharded(smpc, exp_case: int = 12, n_steps: int = 1000, n_pools: int = 8, seed: int = 0,
                            feedback: bool = False, pools=None) -> ClosedLoopReport:
    """The candidate-sharded closed loop (BASELINE config 5 as asked: K = world x K_local, one all-reduce(min) per step) on
    this rank.  ``smpc``: a ``NativeShardedMPC`` -- the whole loop is then ONE library call (``rovmpc_closed_loop_device``
    on the handle that owns the communicators: plant update, rollout, ncclAllReduce and select per step, no host
    synchronisation) -- or a ``ShardedMPC`` (torch.distributed collective; the rehearsal over gloo), for which the same
    plant rule is applied step by step with torch operations.  Every rank draws its own shard of the candidates
    (``closed_loop_pools`` seeded by its k_offset) and ends with the same global records."""
    import torch
    from .sharded import NativeShardedMPC
    engine = smpc.engine
    if isinstance(smpc, NativeShardedMPC):
        return run_closed_loop(engine, exp_case, n_steps, n_pools, seed, smpc.k_offset, feedback, "per_step", pools)
    cfg = engine.cfg
    dev = torch.device("cuda", cfg.device)
    exo_np, state_np = closed_loop_inputs(engine, exp_case, n_steps, seed)
    if pools is None:
        pools = closed_loop_pools(engine, n_pools, seed, smpc.k_offset)
    n_pools = int(pools.shape[0])
    exo = torch.tensor(exo_np, device=dev)
    state = torch.tensor(state_np, device=dev)
    results = torch.empty((n_steps, engine.result_len), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t_start = _time.perf_counter()
    prev = None
    for i in range(n_steps):
        if feedback and prev is not None:
            smpc.join()                                   # the previous global record is produced on the side stream
            th_ga = state[12:14].clone()
            state[0:12] = exo[i, 0:12]; state[14:16] = th_ga; state[12:14] = prev[7:9]
        else:
            state.copy_(exo[i])
        prev = smpc.step_device(state, pools[i % n_pools])
        smpc.join()
        results[i].copy_(prev)                            # (the record buffers rotate)
    smpc.synchronize()
    torch.cuda.synchronize()
    wall = _time.perf_counter() - t_start
    res = results.cpu().numpy()
    tg = np.vstack([res[0, 5:7], res[:, 7:9]])
    sim = n_steps * cfg.dt
    return ClosedLoopReport(n_steps, wall, sim, sim / wall, n_steps * smpc.K_total * cfg.N / wall, res[:, 2:5], tg, res[:, 0], res[:, 1].astype(np.int64))
