#!/usr/bin/env python3
"""BASELINE config 5: closed-loop MPC over a synthetic ROV trajectory (Rov_traj_gen case 12, circular),
N=20, K=4096 candidates per GPU, real-time factor = simulated time / wall time.
Single GPU: python tools/closed_loop_bench.py [--steps 10000]
Multi GPU : torchrun --nproc-per-node G tools/closed_loop_bench.py  (candidate-sharded, RCCL all-reduce(min) per step)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import rovmpc  # noqa: E402
from rovmpc.closed_loop import run_closed_loop  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10000)
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--N", type=int, default=20)
ap.add_argument("--case", type=int, default=12)
args = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
eng = rovmpc.Engine(rovmpc.MPCConfig(N=args.N, K=args.K, device=lr))
k_offset = 0
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
    from rovmpc.sharded import NativeShardedMPC
    smpc = NativeShardedMPC(eng, rank=rank, world=world)
    k_offset = smpc.k_offset
run_closed_loop(eng, args.case, 200, k_offset=k_offset)                      # warm-up
rep = run_closed_loop(eng, args.case, args.steps, k_offset=k_offset)
if rank == 0:
    print(json.dumps({"config": f"closed loop, case {args.case}, {args.steps} steps, N={args.N}, K={args.K} x {world} GPU(s)",
                      "wall_s": rep.wall_s, "sim_s": rep.sim_s, "real_time_factor": rep.real_time_factor,
                      "us_per_step": 1e6 * rep.wall_s / rep.steps, "horizon_steps_per_s": rep.rollouts_per_s * world,
                      "final_cost": float(rep.cost[-1])}))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
