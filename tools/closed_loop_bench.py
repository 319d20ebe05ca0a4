#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: closed-loop MPC over a synthetic ROV trajectory (Rov_traj_gen case 12, circular), N=20,
K=4096, real-time factor = simulated time / wall time, for the two loop forms (launch per step / pipelined),
with the model's own (theta, gamma) fed back and on measured rows.
    python tools/closed_loop_bench.py [--steps 10000] [--short 300]
(--short: a horizon before the fed-back gamma recurrence of the chosen row has drifted off the fast sine path)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import rovmpc  # noqa: E402
from rovmpc.closed_loop import run_closed_loop  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10000)
ap.add_argument("--short", type=int, default=300)
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--N", type=int, default=20)
ap.add_argument("--case", type=int, default=12)
args = ap.parse_args()
eng = rovmpc.Engine(rovmpc.MPCConfig(N=args.N, K=args.K))
out = {"config": f"closed loop, case {args.case}, N={args.N}, K={args.K}, 1 GPU", "runs": []}
for feedback in (True, False):
    for T, reps in ((args.steps, 1), (args.short, 20)):
        for mode in ("per_step", "pipelined"):
            run_closed_loop(eng, args.case, min(T, 200), feedback=feedback, mode=mode)
            walls = [run_closed_loop(eng, args.case, T, feedback=feedback, mode=mode) for _ in range(reps)]
            rep = min(walls, key=lambda r: r.wall_s)
            out["runs"].append({"feedback": feedback, "steps": T, "mode": mode, "us_per_step": 1e6 * rep.wall_s / T,
                                "real_time_factor": rep.real_time_factor, "max_abs_cost": float(np.max(np.abs(rep.cost)))})
print(json.dumps(out))
for r in out["runs"]:
    print(r)
