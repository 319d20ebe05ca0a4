#!/usr/bin/env python3
"""Headline benchmark: MPC horizon-step rollouts/sec at N=20, K=4096 candidates per GPU, fp64
(BASELINE.json configs[1]); candidate-sharded over --gpus ranks with one RCCL all-reduce(min)
per step (configs[3] at 8 GPUs: K=32768 = 8 x 4096).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one full MPC step on device-resident inputs: fused rollout kernel + arg-min epilogue
(+ all-reduce(min) + select when sharded).  Inputs are resident in HBM before the timed region;
a pool of independent synthetic candidate batches is cycled so no step re-reads the previous
step's batch.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (spec), reported as context only


def cpu_baseline(N, seconds_budget=12.0):
    """The oracle ("port" of the reference's per-row Python path) timed on this host, one core:
    reference-style scalar loop (per-row predict, scipy brentq, per-point Rodrigues) on a
    bounded sample of the same workload; the K-vectorised NumPy flavour is reported beside it."""
    import rovmpc
    from oracle import rovmpc_oracle as orc
    model = rovmpc.default_model()
    omodel = orc.DynamicsModel(model.mean, model.scale, orc.SymbolicModel(model.expr_theta),
                               orc.SymbolicModel(model.expr_gamma))
    c = rovmpc.MPCConfig(N=N)
    ocfg = orc.MPCConfig(N=N, dt=c.dt, n_shape_pts=c.n_shape_pts, vt_mode=c.vt_mode, w_T=c.w_T,
                         w_floor=c.w_floor, z_floor=c.z_floor)
    state, U = rovmpc.synthetic_problem(4096, N)
    st = orc.MPCState.from_array(state)
    # scalar flavour: grow the sample until ~seconds_budget is spent
    k, done_units, t_spent = 4, 0, 0.0
    while t_spent < seconds_budget and k <= 4096:
        t0 = time.perf_counter()
        orc.rollout_scalar(ocfg, omodel, st, U[:k])
        dt = time.perf_counter() - t0
        t_spent += dt; done_units += k * N
        if dt * 2 + t_spent > seconds_budget:
            break
        k *= 2
    scalar = done_units / t_spent
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 3.0:
        orc.rollout_vec(ocfg, omodel, st, U)
        reps += 1
    vec = reps * 4096 * N / (time.perf_counter() - t0)
    return {"value": scalar, "unit": "horizon-steps/s", "cores": 1, "kind": "port",
            "sample": f"oracle.rollout_scalar (reference-style per-row loop) on {done_units // N} candidates x N={N}, "
                      f"{t_spent:.1f} s on 1 of {os.cpu_count()} host cores",
            "vectorized_numpy_value": vec,
            "vectorized_sample": f"oracle.rollout_vec, {reps} x (K=4096, N={N}), 1 core"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--N", type=int, default=20)
    ap.add_argument("--K", type=int, default=4096, help="candidates per GPU")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--ck", type=int, default=0, help="candidates per workgroup (0 = auto)")
    ap.add_argument("--nt", type=int, default=0, help="threads per workgroup (0 = auto)")
    ap.add_argument("--pools", type=int, default=8)
    ap.add_argument("--torch-collective", action="store_true", help="use torch.distributed for the all-reduce instead of the library's own RCCL call")
    ap.add_argument("--force-collective", action="store_true",
                    help="single GPU rehearsal of the sharded step: nccl world of 1 with the all-reduce kept")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent MPC steps in flight on one GPU (one engine handle + HIP stream each); 1 = strictly sequential steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-pipelined-extra", action="store_true")
    ap.add_argument("--interp", action="store_true", help="force the bytecode interpreter path")
    ap.add_argument("--model", default="default", help="default | jit-default (reference rows through the hiprtc route) | rows:CT,CG (other Pareto rows) | gen2 | gen3")
    ap.add_argument("--debug-flags", type=int, default=0, help="phase ablation (diagnostics; results invalid)")
    args = ap.parse_args()

    import torch
    import rovmpc
    from rovmpc.sharded import ShardedMPC

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = rovmpc.MPCConfig(N=args.N, K=args.K, dtype=args.dtype, device=local_rank,
                           candidates_per_block=args.ck, threads_per_block=args.nt, force_interpreter=args.interp, debug_flags=args.debug_flags)
    S = max(1, args.streams) if (world == 1 and not args.force_collective) else 1
    model = rovmpc.default_model()
    if args.model == "jit-default":
        model = rovmpc.DynamicsModel(model.mean, model.scale, model.expr_theta + " + 0.0*x0", model.expr_gamma)
    elif args.model == "gen3":          # second-order generation on the features_dd map (dd_cluster.py)
        model = rovmpc.generation3_model()
        cfg.feature_map = rovmpc.FEATURES_GEN3
    elif args.model == "gen2":          # 17 unscaled features (simulate_rk4_theta_gamma.py)
        model = rovmpc.generation2_model()
        cfg.feature_map = rovmpc.FEATURES_GEN2
    elif args.model.startswith("rows:"):
        ct, cg = (int(v) for v in args.model[5:].split(","))
        model = rovmpc.default_model(ct, cg)
    engines = [rovmpc.Engine(cfg, model) for _ in range(S)]
    eng = engines[0]
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    pools = []
    for p in range(args.pools):
        state, U = rovmpc.synthetic_problem(args.K, args.N, seed=20250523 + 1000 * rank + p, dtype=cfg.np_dtype)
        pools.append(torch.tensor(U, device=dev, dtype=tdt).contiguous())
    state, _ = rovmpc.synthetic_problem(1, args.N)          # shared state (same on every rank)
    d_state = torch.tensor(state, device=dev)
    R = eng.result_len
    stream = torch.cuda.current_stream()
    # streams of different priority get different hardware queues (two same-priority streams can share one)
    streams = [stream] if S == 1 else [torch.cuda.Stream(device=dev, priority=-(j % 2)) for j in range(S)]
    smpc = None
    collective = None
    if world > 1 or args.force_collective:
        if not args.torch_collective:
            try:        # RCCL called from the library (one C call per step)
                from rovmpc.sharded import NativeShardedMPC
                smpc = NativeShardedMPC(eng, rank=rank, world=world)
                collective = "ncclAllReduce(min) issued by librovmpc"
            except Exception as exc:                      # noqa: BLE001 -- fall back, never fail the bench
                if rank == 0:
                    print(f"[bench] native RCCL path unavailable ({exc}); using torch.distributed", file=sys.stderr)
                smpc = None
        if smpc is None:
            smpc = ShardedMPC(eng, rank=rank, world=world, force_collective=args.force_collective)
            collective = "torch.distributed all_reduce(MIN) (nccl backend = RCCL)"
    d_res = torch.empty((2 * S, R), dtype=torch.float64, device=dev)

    def one_step(i):
        if smpc is not None:
            return smpc.step_device(d_state, pools[i % args.pools])
        j = i % S
        engines[j].step_device(d_state.data_ptr(), pools[i % args.pools].data_ptr(), d_res[i % (2 * S)].data_ptr(),
                               streams[j].cuda_stream)
        return d_res[i % (2 * S)]

    def fence():
        if smpc is not None:
            smpc.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    fence()
    # HIP events on the launch stream bracket the timed region: with one fused kernel per step,
    # back to back on one stream, (event span) / steps is the average launch-to-launch period of
    # the rollout kernel (its duration plus the ~1.5 us dependent-launch boundary).
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    fence()
    t0 = time.perf_counter()
    for e, st in zip(ev0, streams):
        e.record(st)
    for i in range(args.steps):
        rec = one_step(i)
    for e, st in zip(ev1, streams):
        e.record(st)
    fence()
    elapsed = time.perf_counter() - t0
    # per stream: event span / launches on that stream = launch-to-launch period of the kernel
    launches = [len(range(j, args.steps, S)) for j in range(S)]
    region_ms = sum(a_.elapsed_time(b_) / max(n_, 1) for a_, b_, n_ in zip(ev0, ev1, launches)) / S * args.steps
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    last = rec.cpu().numpy()
    ranks_agree = None
    if dist is not None and world > 1:      # every rank must hold the same global record after the all-reduce
        allrec = [None] * world
        dist.all_gather_object(allrec, last.tolist())
        ranks_agree = all(r == allrec[0] for r in allrec)
    kavg_ms = region_ms / args.steps
    # second, untimed pass: per-launch event pairs around the rollout kernel alone (each pair
    # costs ~3 us of its own, so this pass is not the one `value` comes from)
    kev_ms = kev_min_ms = None
    if not args.no_kernel_timing:
        n_ev = min(args.steps, 200)
        eng.timing_enable(n_ev)
        for i in range(n_ev * S):
            one_step(i)
        fence()
        kev_ms, kev_min_ms, _ = eng.timing_read()
        eng.timing_enable(0)

    if rank == 0:
        units_per_step = world * args.K * args.N
        tag = ("C2" if (args.N, args.K, args.dtype) == (20, 4096, "f64") else
               "C3" if (args.N, args.K, args.dtype) == (50, 16384, "f32") else "other size")     # BASELINE.json configs
        esz = 8 if args.dtype == "f64" else 4
        alg_bytes = args.K * args.N * 3 * esz + args.K * esz          # controls in, costs out
        out = {
            "metric": "MPC rollouts/sec (horizon-steps/sec) at N=20, K=4096; 1/2/4/8 GPU",
            "value": units_per_step * args.steps / elapsed,
            "unit": "horizon-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{tag}: one MPC step, N={args.N} horizon x K={args.K} candidates per GPU "
                                   f"(global K={world * args.K}), fused RK4+catenary HIP kernel, {args.dtype}",
                       "N": args.N, "K_per_gpu": args.K, "K_global": world * args.K,
                       "n_shape_pts": cfg.n_shape_pts, "vt_mode": "compose", "dt": cfg.dt,
                       "model": f"{args.model}: {eng.model_path}",
                       "parallelism": f"candidate-sharded x{world}, 1 all-reduce(min)/step" if world > 1 else "single GPU",
                       "steps_in_flight": S if smpc is None else "rollout(i+1) overlaps all-reduce(i)",
                       "collective": collective,
                       "candidates_per_workgroup": eng.cfg.candidates_per_block or "auto"},
            "best": {"cost": float(last[0]), "index": int(last[1])},
            "ranks_agree": ranks_agree,
        }
        if kavg_ms:
            achieved = alg_bytes / (kavg_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "rollout_kernel", "kernel_avg_us": kavg_ms * 1e3,
                               "chip_algorithmic_GBps": alg_bytes * args.steps / elapsed / 1e9,
                               "kernel_event_pair_us": kev_ms * 1e3 if kev_ms else None,
                               "kernel_event_pair_min_us": kev_min_ms * 1e3 if kev_min_ms else None,
                               "algorithmic_bytes_per_launch": alg_bytes,
                               "note": "fp64 VALU/latency-bound by construction (~2-3 kFLOP of transcendental work "
                                       "per 24.4 B); see DESIGN.md"}
        # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # separate runs of this same command; MI355X_MICROARCH.md: KiB units, FETCH_SIZE reads half the
        # bytes of a 16 B/lane coalesced stream on gfx950 -> doubled); only for the profiled workload.
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if "roofline" in out and os.path.exists(pmc_path) and (args.N, args.K, args.dtype, world) == (20, 4096, "f64", 1):
            pmc = json.load(open(pmc_path))
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                out["roofline"]["traffic"] = (2.0 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024.0
                out["roofline"]["traffic_source"] = "profiles/r01_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE, KiB)"
            if "SQ_WAVE_CYCLES" in pmc and "SQ_WAIT_ANY" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
                # what actually bounds the kernel (same committed PMC passes): share of the waves' lifetime spent
                # waiting (barriers, dependent latencies) vs issuing VALU
                wc = pmc["SQ_WAVE_CYCLES"]["mean"]
                out["roofline"]["wave_cycles_waiting_frac"] = pmc["SQ_WAIT_ANY"]["mean"] / wc
                out["roofline"]["wave_cycles_valu_frac"] = pmc["SQ_ACTIVE_INST_VALU"]["mean"] / wc
                if "SQ_INSTS_VALU" in pmc:
                    out["roofline"]["valu_wave_instructions_per_launch"] = pmc["SQ_INSTS_VALU"]["mean"]
        if world == 1 and smpc is None and S == 1 and not args.no_pipelined_extra:
            # extra, not the headline: two independent MPC steps in flight on one GPU (second engine handle
            # on a high-priority stream = its own hardware queue), so one step's launch ramp / arg-min tail
            # overlaps the other's work
            eng2 = rovmpc.Engine(cfg, model)
            st2 = torch.cuda.Stream(device=dev, priority=-1)
            pair = [(eng, stream), (eng2, st2)]
            r2 = torch.empty((4, R), dtype=torch.float64, device=dev)
            def step2(i):
                e, st = pair[i & 1]
                e.step_device(d_state.data_ptr(), pools[i % args.pools].data_ptr(), r2[i & 3].data_ptr(), st.cuda_stream)
            for i in range(args.warmup):
                step2(i)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(args.steps):
                step2(i)
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t2
            out["two_steps_in_flight"] = {"value": units_per_step * args.steps / e2, "ms_per_step": 1e3 * e2 / args.steps,
                                          "note": "throughput with 2 independent steps overlapped on one GPU; `value` above is 1 in flight"}
            eng2.close()
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.N)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for e in engines:
        e.close()


if __name__ == "__main__":
    main()
