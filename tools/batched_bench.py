#!/usr/bin/env python3
"""B independent C2-sized problems per launch (rovmpc_step_batch_device) -- the throughput regime of the rollout kernel --
as a stand-alone command, so that rocprofv3 can take its kernel trace and PMC passes on exactly this launch.

    python3 tools/batched_bench.py [--B 64 --N 20 --K 4096 --dtype f64 --steps 100 --warmup 10]

Prints one JSON line: horizon-steps/s, ms per launch (wall, synchronised at both ends) and the HIP-event span per launch.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--N", type=int, default=20)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ck", type=int, default=0)
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--debug-flags", type=int, default=0, help="phase ablation (diagnostics; results invalid): 1 no integration, 2 no geometry, 4 no phase 2")
    ap.add_argument("--check", action="store_true", help="compare problems 0 and B-1 with their single launches, bit for bit")
    args = ap.parse_args()
    import torch
    import rovmpc
    dev = torch.device("cuda", 0)
    cfg = rovmpc.MPCConfig(N=args.N, K=args.K, dtype=args.dtype, candidates_per_block=args.ck, threads_per_block=args.nt,
                           debug_flags=args.debug_flags)
    B, N, K = args.B, args.N, args.K
    with rovmpc.Engine(cfg) as eng:
        R = eng.result_len
        states = np.empty((B, 16)); U = np.empty((B, K, N, 3), dtype=cfg.np_dtype)
        for b in range(B):
            states[b], U[b] = rovmpc.synthetic_problem(K, N, seed=777 + b, dtype=cfg.np_dtype)
        d_states = torch.tensor(states, device=dev); d_U = torch.tensor(U, device=dev)
        d_res = torch.empty((B, R), dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        p_s, p_u, p_r = d_states.data_ptr(), d_U.data_ptr(), d_res.data_ptr()
        for _ in range(args.warmup):
            eng.step_batch_device(B, p_s, p_u, p_r, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            eng.step_batch_device(B, p_s, p_u, p_r, stream)
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        same = None
        if args.check:
            single = torch.empty(R, dtype=torch.float64, device=dev)
            same = True
            for b in (0, B - 1):
                eng.step_device(d_states[b].data_ptr(), d_U[b].data_ptr(), single.data_ptr(), stream)
                torch.cuda.synchronize()
                same = same and bool(torch.equal(single, d_res[b]))
        esz = 8 if args.dtype == "f64" else 4
        alg = B * (K * N * 3 * esz + K * esz)
        ev_ms = e0.elapsed_time(e1) / args.steps
        print(json.dumps({"workload": f"batched: B={B} problems x (N={N}, K={K}, {args.dtype}) per launch",
                          "value": B * K * N * args.steps / wall, "unit": "horizon-steps/s",
                          "ms_per_launch": 1e3 * wall / args.steps, "event_ms_per_launch": ev_ms,
                          "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": alg / (ev_ms * 1e-3) / 1e9,
                          "records_bit_equal_to_single_launches": same}))


if __name__ == "__main__":
    main()
