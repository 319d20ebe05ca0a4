"""Synthetic two-ROV trajectories: the 14 experiment cases of Rov_traj_gen.py:7-116,
parameterised on n_steps / total_time, with a seeded RNG for the PRBS cases 9 and 10
(the reference draws those from the unseeded global np.random)."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

HEADER = ("br0_x, br0_y, br0_z, br0_phi, br0_theta, br0_psi, br0_u, br0_v, br0_w, br0_p, br0_q, br0_r, "
          "br1_x, br1_y, br1_z, br1_phi, br1_theta, br1_psi, br1_u, br1_v, br1_w, br1_p, br1_q, br1_r")


def generate_rov_trajectories(exp_case: int, n_steps: int = 100, total_time: float = 10.0, separation: float = 1.0,
                              seed: Optional[int] = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(time (n_steps,), trajectory_0 (12,n_steps), trajectory_1 (12,n_steps)); rows = x y z phi
    theta psi u v w p q r."""
    if not 1 <= exp_case <= 14:
        raise ValueError("exp_case must be 1..14")
    t = np.linspace(0, total_time, n_steps)
    a = np.zeros((12, n_steps)); b = np.zeros((12, n_steps))
    rng = np.random.default_rng(seed)
    w = 2 * np.pi * t
    lin = {1: (0.03, 0.03), 2: (0.03, 0.06), 3: (0.03, -0.03), 13: (0.06, 0.06)}
    if exp_case in lin:                                   # straight runs, cases 1-3 and 13
        va, vb = lin[exp_case]
        a[0], b[0], a[6], b[6] = va * t, vb * t, va, vb
        b[1] = separation
    elif exp_case == 4:                                   # one static, one moving
        b[0] = 0.05 * t; b[1] = separation; b[6] = 0.5
    elif exp_case in (5, 6):                              # depth variation while advancing
        vb = 0.03 if exp_case == 5 else 0.06
        a[0], b[0], a[6], b[6] = 0.03 * t, vb * t, 0.03, vb
        b[1] = separation; a[2] = 0.5; b[2] = np.linspace(0.5, 1.0, n_steps)
    elif exp_case == 7:                                   # depth variation, ROV 1 static
        b[1] = separation; a[2] = 0.5; b[2] = np.linspace(0.5, 1.0, n_steps); b[6] = 0.05
    elif exp_case == 8:                                   # rapid lateral oscillation
        a[0] = b[0] = 0.05 * t
        a[1] = 0.05 * np.sin(w); b[1] = separation + 0.05 * np.sin(w)
        a[6] = b[6] = 0.05 * np.cos(w / total_time)
    elif exp_case == 9:                                   # PRBS on ROV 1
        a[0] = rng.choice([-0.1, 0.1], n_steps); b[0] = 0.05 * t; b[1] = separation
        a[6] = rng.choice([-0.03, 0.03], n_steps)
    elif exp_case == 10:                                  # PRBS on both
        a[0] = rng.choice([-0.1, 0.1], n_steps); b[0] = rng.choice([-0.1, 0.1], n_steps); b[1] = separation
        a[6] = rng.choice([-0.03, 0.03], n_steps); b[6] = rng.choice([-0.03, 0.03], n_steps)
    elif exp_case == 11:                                  # zig-zag
        a[0] = b[0] = 0.05 * t; b[1] = separation; a[1] = 0.2 * np.sin(w); b[6] = 0.03
    elif exp_case == 12:                                  # concentric circles
        a[0], a[1] = 0.4 * np.cos(w / total_time), 0.4 * np.sin(w / total_time)
        b[0], b[1] = 0.1 * np.cos(w / total_time), 0.1 * np.sin(w / total_time)
    elif exp_case == 14:                                  # static cable drift
        b[1] = separation
    return t, a, b


def trajectory_csv(traj0: np.ndarray, traj1: np.ndarray) -> str:
    """CSV text in the reference's layout (header + ``%.3f`` rows, Rov_traj_gen.py:131-139)."""
    lines = [HEADER]
    for s0, s1 in zip(traj0.T, traj1.T):
        lines.append(",".join(f"{v:.3f}" for v in s0) + "," + ",".join(f"{v:.3f}" for v in s1))
    return "\n".join(lines) + "\n"
