#!/usr/bin/env python3
"""Headline benchmark: MPC horizon-step rollouts/sec at N=20, K=4096 candidates per GPU, fp64
(BASELINE.json configs[1]); candidate-sharded over --gpus ranks with one RCCL all-reduce(min)
per step (configs[3] at 8 GPUs: K=32768 = 8 x 4096).

    python bench.py [--gpus N --steps K --warmup W]

`--gpus N` with N > 1 is a plain command: when no rank environment is present the process
starts N rank processes itself (`python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a CHILD, before anything here touches the GPU), relays rank 0's JSON line and exits with the
children's status.  Launched under torch.distributed.run directly (RANK/WORLD_SIZE set) it is a rank.

A "step" = one full MPC step on device-resident inputs: fused rollout kernel + arg-min epilogue
(+ all-reduce(min) + select when sharded).  Inputs are resident in HBM before the timed region;
a pool of independent synthetic candidate batches is cycled so no step re-reads the previous
step's batch.  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "catenary-model-estimation-and-mpc-control-for-rov-tethered-systems_amd")

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (spec): 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz
PROFILE_TAG = "r03"             # profiles/<tag>_pmc_summary.json is the PMC pass the roofline block may quote


def kernel_sources_sha16():
    """Hash of the sources the rollout kernel is built from; stamped into every profile summary so a
    committed PMC pass is only quoted for the kernel it was taken on."""
    h = hashlib.sha256()
    for rel in ("csrc/rollout_kernels.h", "csrc/device_math.h", "csrc/rovmpc.hip", "csrc/util_kernels.h"):
        with open(os.path.join(PKG, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--presteps", type=int, default=-1,
                    help="further untimed steps in front of --warmup (clocks and caches settle over the first ~200 launches); "
                         "-1 = as many as bring the untimed steps to 200, 0 = none; reported as untimed_presteps")
    ap.add_argument("--N", type=int, default=20)
    ap.add_argument("--K", type=int, default=4096, help="candidates per GPU")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--dt", type=float, default=0.0, help="horizon step [s] (0 = the library default, 1/60)")
    ap.add_argument("--ck", type=int, default=0, help="candidates per workgroup (0 = auto)")
    ap.add_argument("--nt", type=int, default=0, help="threads per workgroup (0 = auto)")
    ap.add_argument("--pools", type=int, default=8)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the ranks; gloo = rehearsal of the N > 1 path without RCCL "
                         "(the slot buffer crosses the host)")
    ap.add_argument("--devices", default="", help="comma list: device ordinal of each local rank (default: the local rank); "
                                                  "'0,0' rehearses two ranks on one GPU (gloo only: RCCL refuses a shared GPU)")
    ap.add_argument("--protocol-only", action="store_true",
                    help="no GPU, no rollout: every rank fabricates its record; exercises launcher, rendezvous, "
                         "pack / all-reduce(min) / select and the JSON relay (CPU test of the N > 1 plumbing)")
    ap.add_argument("--torch-collective", action="store_true", help="use torch.distributed for the all-reduce instead of the library's own RCCL call")
    ap.add_argument("--force-collective", action="store_true",
                    help="single GPU rehearsal of the sharded step: nccl world of 1 with the all-reduce kept")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent MPC steps in flight on one GPU (one engine handle + HIP stream each); 1 = strictly sequential steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-pipelined-extra", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip batched / worst-path / dt=0.05 / closed-loop extras")
    ap.add_argument("--closed-loop", type=int, default=-1,
                    help="steps of the closed-loop extra (BASELINE config 5, Rov_traj_gen case 12); -1 = 10000 at the default size, 0 = off")
    ap.add_argument("--interp", action="store_true", help="force the bytecode interpreter path")
    ap.add_argument("--model", default="default", help="default | jit-default (reference rows through the hiprtc route) | rows:CT,CG (other Pareto rows) | gen2 | gen3")
    ap.add_argument("--debug-flags", type=int, default=0, help="phase ablation (diagnostics; results invalid)")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="seconds the self-launched ranks may take")
    ap.add_argument("--native-check-timeout", type=float, default=90.0,
                    help="seconds the check of the library's own RCCL path may take at N > 1 before its communicators are aborted "
                         "and the run continues on torch.distributed's collective")
    ap.add_argument("--fallback-reason", default="", help=argparse.SUPPRESS)   # set by the parent when it re-launches
    return ap


# ---- parent: start the ranks as children (never exec, never touch the GPU here) --------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(argv, n, timeout):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        try:                                     # exactly the process group started above
            os.killpg(proc.pid, signal.SIGTERM)
            time.sleep(5)
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, _ = proc.communicate()
        rc = 124
    line = None
    for ln in (out or "").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
    return rc, line, out or ""


def launch_ranks(args, argv):
    native = not args.torch_collective and not args.protocol_only and args.backend == "nccl"
    # the library's own RCCL path has never run at world > 1 (RCCL refuses two ranks on the one GPU of the build box):
    # it gets a bounded first attempt; if it dies OR does not finish, its whole process group is killed and the run is
    # repeated once on torch.distributed's collective, with the reason carried into the JSON line -- never a silent switch
    # (two native modes are checked in-run, each cut short after --native-check-timeout seconds; the timed passes and the
    # sharded closed loop follow)
    t_first = min(args.launch_timeout, 2.0 * args.native_check_timeout + 240.0) if native else args.launch_timeout
    t0 = time.time()
    rc, line, out = _run_ranks(argv, args.gpus, t_first)
    if (rc != 0 or line is None) and native:
        why = (f"native RCCL run did not finish within {t_first:.0f} s" if rc == 124 else f"native RCCL run exited with code {rc}") + \
              ("" if line else " without a result line")
        print(f"[bench] {why}; re-running with --torch-collective", file=sys.stderr)
        left = max(120.0, args.launch_timeout - (time.time() - t0))
        rc, line, out = _run_ranks(argv + ["--torch-collective", "--fallback-reason", why], args.gpus, left)
    if line is None:
        sys.stderr.write(out[-4000:])
        print(f"[bench] ranks produced no result line (exit code {rc})", file=sys.stderr)
        sys.exit(rc or 1)
    print(line)
    sys.exit(rc)


# ---- CPU baseline -----------------------------------------------------------------------------------------------------

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


# ---- protocol-only ranks (no GPU): the N > 1 plumbing under gloo -------------------------------------------------------

def protocol_rank(args, world, rank):
    """Every rank fabricates a record per step (cost and index a fixed function of (rank, step)), packs it into its row of
    the [world][R] int64 slot image, ONE all-reduce(min), select -- the host statement of the sharded step.  Checks that
    every rank ends with the same record and that it is the lexicographic minimum; prints the same JSON shape."""
    import torch
    import torch.distributed as dist
    from rovmpc.sharded import ShardedMPC
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = 5 + 2 * (args.N + 1)
    step_no = [0]

    def fabricated(r, i):
        g = np.random.default_rng(1000 * i + r)
        rec = g.standard_normal(R)
        rec[0] = float((7 * i + 3 * r) % world) + 0.25          # ties across ranks happen: the lower index must win
        rec[1] = float(r * args.K + int(g.integers(0, args.K)))
        return rec

    smpc = ShardedMPC(local_solver=lambda: torch.tensor(fabricated(rank, step_no[0])), rank=rank, world=world,
                      K_total=world * args.K)
    ok = True
    for i in range(args.warmup):
        step_no[0] = i; smpc.step_host()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step_no[0] = i
        rec = smpc.step_host().numpy()
        want = min((fabricated(r, i) for r in range(world)), key=lambda v: (v[0], v[1]))
        ok = ok and np.array_equal(rec, want)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    allrec = [None] * world
    dist.all_gather_object(allrec, (rec.tolist(), bool(ok)))
    if rank == 0:
        print(json.dumps({
            "metric": metric_name(args, world), "value": None, "unit": "horizon-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * float(t.item()) / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "protocol-only: fabricated per-rank records, no rollout (plumbing check, not a measurement)",
            "config": {"workload": "protocol-only", "N": args.N, "K_per_gpu": args.K, "K_global": world * args.K,
                       "collective": "torch.distributed all_reduce(MIN) (gloo)", "collective_fallback_reason": None},
            "ranks_agree": all(r[0] == allrec[0][0] for r in allrec),
            "records_are_the_global_min": all(r[1] for r in allrec)}))
    dist.barrier()
    dist.destroy_process_group()


def check_native_path(args, eng, smpc, d_state, pools, rank, world, dev, dist):
    """64 steps through the library's own RCCL path, the last of them also through torch.distributed's collective (same
    kernels, same slot image): the two global records must be the same bits on every rank and no hand-off may have timed
    out.  Returns None when the path may be timed, else the reason (agreed across ranks with one all-reduce).  A collective
    that never completes is cut short: the GPU-side hand-off waits run on a 500 ms clock during the check, the error word is
    polled between steps (a stuck collective then costs one give-up, not one per remaining step), and a timer thread calls
    rovmpc_comm_abort after --native-check-timeout seconds so that the blocked synchronise returns its error."""
    import threading
    import torch
    from rovmpc.sharded import ShardedMPC
    why, fired, got = None, [], None

    def _abort():
        fired.append(True)
        try:
            eng.comm_abort()
        except Exception:                                 # noqa: BLE001 -- the check reports the time-out either way
            pass
    eng.set_option("handoff_timeout_ms", 500.0)
    timer = threading.Timer(args.native_check_timeout, _abort)
    timer.daemon = True
    timer.start()
    try:
        if os.environ.get("ROVMPC_BENCH_TEST_VALIDATE") == "abort":     # test hook: as if the check had timed out
            _abort()
        for i in range(64):                               # many turns of the slot and communicator rotations
            got = smpc.step_device(d_state, pools[i % args.pools])
            if i % 8 == 7:
                eng.device_status()                       # a hand-off already gave up: stop here
        smpc.synchronize()
        got = got.cpu().numpy()
    except Exception as exc:                              # noqa: BLE001 -- recorded in the JSON line
        why, got = f"native RCCL path failed its check: {exc}", None
    finally:
        timer.cancel()
    if fired:
        why = (f"native RCCL path did not finish its check within {args.native_check_timeout:.0f} s; its communicators "
               f"were aborted" + (f" ({why})" if why else ""))
        got = None
    eng.set_option("handoff_timeout_ms", 10000.0)
    ref = ShardedMPC(eng, rank=rank, world=world, force_collective=args.force_collective)
    want = ref.step_device(d_state, pools[63 % args.pools])
    ref.synchronize()
    want = want.cpu().numpy()
    del ref
    if os.environ.get("ROVMPC_BENCH_TEST_VALIDATE") == "fail":          # test hook: the check fails on this rank
        got = None if got is None else got + 1.0
    if why is None and (got is None or got.tobytes() != want.tobytes()):
        why = "native RCCL path failed its check: global record differs from the torch.distributed collective's"
    flag = torch.tensor([1 if why else 0], dtype=torch.int32, device=dev)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) and why is None:
        why = "native RCCL path failed its check on another rank"
    return why


def metric_name(args, world):
    if (args.N, args.K) == (20, 4096):
        return "MPC rollouts/sec (horizon-steps/sec) at N=20, K=4096; 1/2/4/8 GPU"      # BASELINE.json's metric, verbatim
    return f"MPC rollouts/sec (horizon-steps/sec) at N={args.N}, K={args.K}; {world} GPU"


# ---- a rank -------------------------------------------------------------------------------------------------------------

def main():
    args = build_parser().parse_args()
    have_rank_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_rank_env:
        launch_ranks(args, sys.argv[1:])          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.protocol_only:
        if os.environ.get("ROVMPC_BENCH_TEST_FAIL_RANK") == str(rank):      # test hook: this rank dies before the rendezvous
            raise SystemExit(3)
        return protocol_rank(args, world, rank)

    if world > 1 or args.force_collective:
        # the sharded step keeps up to five streams busy (rollouts, three collective streams, torch's own): let the runtime
        # open a hardware queue for each instead of folding them onto its default four (read at HIP start-up, so before torch)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import rovmpc
    from rovmpc.sharded import ShardedMPC

    devs = [int(v) for v in args.devices.split(",")] if args.devices else None
    device = devs[local_rank] if devs else local_rank
    if devs and len(set(devs)) < len(devs) and args.backend == "nccl" and world > 1:
        raise SystemExit("--devices maps two ranks to one GPU: RCCL refuses that (Duplicate GPU detected); use --backend gloo")
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    cfg = rovmpc.MPCConfig(N=args.N, K=args.K, dtype=args.dtype, device=device,
                           candidates_per_block=args.ck, threads_per_block=args.nt, force_interpreter=args.interp, debug_flags=args.debug_flags)
    if args.dt > 0:
        cfg.dt = args.dt
    S = max(1, args.streams) if (world == 1 and not args.force_collective) else 1
    model = rovmpc.default_model()
    if args.model == "jit-default":
        cfg.no_builtin = True               # the reference rows through the hiprtc route
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
    eng = eng0 = engines[0]
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
    fallback_reason = args.fallback_reason or None
    native = {}                      # communicators -> (engine, NativeShardedMPC) that passed the in-run check
    native_why = {}                  # communicators -> why that mode is not used
    if world > 1 or args.force_collective:
        if not args.torch_collective and args.backend == "nccl":
            # The library's own RCCL path cannot run at world > 1 on the one-GPU build box, so every multi-rank run CHECKS it
            # before timing it, one mode after the other: three communicators in rotation (the collectives of consecutive
            # steps overlap each other), then the conservative single communicator.  A mode is used only if its check passes
            # on every rank; `value` is the three-communicator timing when that mode passed, else the single-communicator one,
            # and both are printed (extra.multi_comm_value / extra.single_comm_value).  No mode passes: torch.distributed's
            # collective, with the reasons in the line and value_is_fallback = true.
            from rovmpc.sharded import NativeShardedMPC
            want_modes = [int(v) for v in os.environ.get("ROVMPC_BENCH_COMM_MODES", "3,1").split(",")]
            for comms in want_modes:
                e_n = eng if not native and comms == want_modes[0] else rovmpc.Engine(cfg, model)
                os.environ["ROVMPC_COMMS"] = str(comms)          # read by rovmpc_comm_init
                why = None
                try:
                    s_n = NativeShardedMPC(e_n, rank=rank, world=world)
                except Exception as exc:                      # noqa: BLE001 -- recorded in the JSON line
                    why, s_n = f"native RCCL set-up failed: {exc}", None
                if s_n is not None and (world > 1 or os.environ.get("ROVMPC_BENCH_TEST_VALIDATE")):
                    why = check_native_path(args, e_n, s_n, d_state, pools, rank, world, dev, dist)
                if why is None:
                    native[comms] = (e_n, s_n)
                else:
                    native_why[comms] = why
                    if rank == 0:
                        print(f"[bench] {comms}-communicator mode not used: {why}", file=sys.stderr)
                    if s_n is not None:
                        try:
                            s_n.close()
                        except Exception:                     # noqa: BLE001 -- the communicator is being abandoned anyway
                            pass
                    if e_n is not eng:
                        e_n.close()
            os.environ.pop("ROVMPC_COMMS", None)
            if native:
                head = max(native)                            # most communicators that passed
                eng_h, smpc = native[head]
                collective = f"ncclAllReduce(min) issued by librovmpc ({head} communicator(s) in rotation, 4 steps in flight)"
                if eng_h is not eng:
                    engines = [eng_h]; eng = eng_h
            else:
                fallback_reason = "; ".join(f"{c} communicator(s): {w}" for c, w in native_why.items()) or fallback_reason
        if smpc is None:
            smpc = ShardedMPC(eng, rank=rank, world=world, force_collective=args.force_collective,
                              host_staged=(args.backend == "gloo"))
            collective = ("torch.distributed all_reduce(MIN) (nccl backend = RCCL)" if args.backend == "nccl" else
                          "torch.distributed all_reduce(MIN) (gloo, slot image staged through the host: rehearsal)")
    d_res = torch.empty((2 * S, R), dtype=torch.float64, device=dev)
    hang_guard = None
    if world > 1 and collective and collective.startswith("ncclAllReduce"):
        # the checked native path goes on into the measurement; should a collective still never complete there, the run must
        # end as a failure with a reason, not as a hang: after --launch-timeout / 2 seconds the communicators are aborted
        # and the blocked synchronise raises
        import threading
        hang_guard = threading.Timer(max(60.0, args.launch_timeout / 2), eng.comm_abort)
        hang_guard.daemon = True
        hang_guard.start()

    # the launch arguments of the plain step are plain integers, taken once: the host side of a timed launch is one ctypes
    # call (a tensor index + data_ptr() per argument costs more than the call)
    p_state, p_U = d_state.data_ptr(), [u.data_ptr() for u in pools]
    p_res, p_streams = [d_res[r].data_ptr() for r in range(2 * S)], [st.cuda_stream for st in streams]
    f_step, n_pools = [e.step_device for e in engines], args.pools

    def one_step(i, sm=None):
        sm = sm if sm is not None else smpc
        if sm is not None:
            return sm.step_device(d_state, pools[i % n_pools])
        f_step[i % S](p_state, p_U[i % n_pools], p_res[i % (2 * S)], p_streams[i % S])
        return i % (2 * S)                    # row of d_res

    def fence(sm=None):
        sm = sm if sm is not None else smpc
        if sm is not None:
            sm.synchronize()                  # raises if a GPU-side hand-off of any step gave up
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # clocks and caches settle over the first few hundred launches (measured: the 20 steps behind 5 warm-ups run 5 % slower
    # than steady state); --presteps further UNTIMED steps are run in front of --warmup and reported as such (default: as
    # many as bring the untimed steps to 200; --presteps 0 switches them off)
    presteps = 0 if args.protocol_only else (max(0, 200 - args.warmup) if args.presteps < 0 else args.presteps)
    # HIP events on the launch stream bracket the timed region: with one fused kernel per step,
    # back to back on one stream, (event span) / steps is the average launch-to-launch period of
    # the rollout kernel (its duration plus the ~1.5 us dependent-launch boundary).
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    for i in range(presteps + args.warmup):
        one_step(i)
    if smpc is None:
        # drain the untimed steps SPINNING: a host thread that blocks for the milliseconds they take comes back from a deep
        # sleep state and feeds the first timed launches late (measured: +2 us per step over a 20-step region)
        ev_w = torch.cuda.Event()
        ev_w.record(streams[-1])
        while not ev_w.query():
            pass
    if dist is not None:
        fence(); fence()                      # (the collective library's barrier is slow the first times it runs: 139 us, then 35)
    fence()                                   # barrier + synchronise: the GPU idles only for this one round trip
    t0 = time.perf_counter()
    for e, st in zip(ev0, streams):
        e.record(st)
    _ta = time.perf_counter()
    for i in range(args.steps):
        rec = one_step(i)
    _tb = time.perf_counter()
    for e, st in zip(ev1, streams):
        e.record(st)
    _tc = time.perf_counter()
    if smpc is None:
        # spin on the end events before the blocking synchronise below: a blocked host thread is woken tens of
        # microseconds after the GPU finishes, which is 5 % of a 20-step timed region (the synchronise still brackets it)
        while not all(e.query() for e in ev1):
            pass
    _td = time.perf_counter()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if os.environ.get("ROVMPC_BENCH_TRACE_BRACKET"):
        print("[bracket] record0 %.1f us, launches %.1f, record1 %.1f, spin %.1f, fence %.1f, total %.1f" % tuple(
            1e6 * x for x in (_ta - t0, _tb - _ta, _tc - _tb, _td - _tc, time.perf_counter() - _td, time.perf_counter() - t0)), file=sys.stderr)
    # per stream: event span / launches on that stream = launch-to-launch period of the kernel
    launches = [len(range(j, args.steps, S)) for j in range(S)]
    region_ms = sum(a_.elapsed_time(b_) / max(n_, 1) for a_, b_, n_ in zip(ev0, ev1, launches)) / S * args.steps
    last = (rec if smpc is not None else d_res[rec]).cpu().numpy()
    ranks_agree = None
    if dist is not None and world > 1:      # every rank must hold the same global record after the all-reduce
        allrec = [None] * world
        dist.all_gather_object(allrec, last.tolist())
        ranks_agree = all(r == allrec[0] for r in allrec)

    # Sharded step: the strict bracket above ends with the drain of a four-step pipeline, a synchronise across seven queues
    # and the collective library's barrier -- ~170 us that 20 timed steps of ~22 us cannot amortise.  The
    # sharded `value` is therefore the pipeline's steady state: K steps between two HIP events on the caller's stream with
    # FILL untimed steps in flight in front of the first event and behind the second (the pipeline is as full when the
    # region ends as when it starts, so exactly K steps' worth of work completes inside it), barrier + synchronise outside;
    # MAX over ranks.  The strict bracket is printed next to it (`strict_bracket`).
    steady = {}
    FILL = 4

    def steady_state(sm):
        cur = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence(sm)
        for i in range(FILL):
            one_step(i, sm)
        e0.record(cur)
        for i in range(args.steps):
            one_step(FILL + i, sm)
        e1.record(cur)
        for i in range(FILL):
            one_step(FILL + args.steps + i, sm)
        fence(sm)
        return max_over_ranks(e0.elapsed_time(e1) * 1e-3)

    if smpc is not None:
        steady["headline"] = steady_state(smpc)
        for comms, (e_n, s_n) in native.items():
            if s_n is not smpc:
                for i in range(presteps + args.warmup):
                    one_step(i, s_n)
                steady[comms] = steady_state(s_n)
            else:
                steady[comms] = steady["headline"]
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

    # BASELINE config 5 in its sharded form: the closed loop of Rov_traj_gen case 12 with the model's own (theta, gamma) fed
    # back, every step the candidate-sharded one (rollout + all-reduce(min) + select); every rank runs it, rank 0 reports
    closed_loop_sharded = None
    if smpc is not None and args.closed_loop != 0:
        import hashlib as _hl
        from rovmpc.closed_loop import run_closed_loop_sharded
        native_loop = (collective or "").startswith("ncclAllReduce")
        T = args.closed_loop if args.closed_loop > 0 else (10000 if native_loop else 200)
        try:
            run_closed_loop_sharded(smpc, 12, min(T, 100), feedback=True)                      # warm-up
            fence()
            t_cl = time.perf_counter()
            rep = run_closed_loop_sharded(smpc, 12, T, feedback=True)
            fence()
            wall_cl = max_over_ranks(time.perf_counter() - t_cl)
            digest = _hl.sha256(rep.cost.tobytes() + rep.index.tobytes() + rep.u.tobytes() + rep.theta_gamma.tobytes()).hexdigest()
            agree = None
            if dist is not None and world > 1:
                alld = [None] * world
                dist.all_gather_object(alld, digest)
                agree = all(d_ == alld[0] for d_ in alld)
            closed_loop_sharded = {"steps": T, "case": 12, "feedback": True, "n_gpus": world, "K_global": world * args.K, "N": args.N,
                                   "us_per_step": 1e6 * wall_cl / T, "real_time_factor": T * cfg.dt / wall_cl,
                                   "ranks_agree": agree, "final_cost": float(rep.cost[-1]),
                                   "all_costs_finite": bool(np.isfinite(rep.cost).all()),
                                   "path": ("rovmpc_closed_loop_device on the handle that owns the communicators: plant update, rollout, "
                                            "ncclAllReduce(min), select per step, one library call for the whole loop") if native_loop else
                                           "python loop over ShardedMPC.step_device (torch.distributed collective) with the same plant rule",
                                   "note": "a true closed loop cannot hide the collective: step = plant update + rollout + all-reduce + select"}
        except Exception as exc:                              # noqa: BLE001 -- reported in the line
            closed_loop_sharded = {"error": str(exc)}

    if hang_guard is not None:
        hang_guard.cancel()
    if rank == 0:
        import bench_extras
        units_per_step = world * args.K * args.N
        strict_elapsed = elapsed
        timing_note = "strict bracket: K steps between barrier + synchronise, wall clock, MAX over ranks"
        if smpc is not None and steady.get("headline"):
            elapsed = steady["headline"]
            timing_note = (f"steady state of the sharded pipeline: HIP-event span of K steps on the caller's stream with {FILL} untimed "
                           f"steps in flight before the first event and after the second, barrier + synchronise outside, MAX over "
                           f"ranks; the strict bracket (drain + synchronise + barrier inside the timed region) is `strict_bracket`")
        tag = ("C2" if (args.N, args.K, args.dtype) == (20, 4096, "f64") else
               "C3" if (args.N, args.K, args.dtype) == (50, 16384, "f32") else "other size")     # BASELINE.json configs
        esz = 8 if args.dtype == "f64" else 4
        alg_bytes = args.K * args.N * 3 * esz + args.K * esz          # controls in, costs out
        sha = kernel_sources_sha16()
        out = {
            "metric": metric_name(args, world),
            "value": units_per_step * args.steps / elapsed,
            "unit": "horizon-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_presteps": presteps,
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
                       "collective": collective, "collective_fallback_reason": fallback_reason,
                       "backend": args.backend if dist is not None else None,
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "comm_placement": (eng.comm_placement() or None) if (collective or "").startswith("ncclAllReduce") else None,
                       "devices": devs, "candidates_per_workgroup": eng.cfg.candidates_per_block or "auto"},
            "best": {"cost": float(last[0]), "index": int(last[1])},
            "ranks_agree": ranks_agree,
            "build": {"kernel_sources_sha16": sha},
            "timing": timing_note,
        }
        if smpc is not None:
            native_head = (collective or "").startswith("ncclAllReduce")
            out["strict_bracket"] = {"value": units_per_step * args.steps / strict_elapsed, "ms_per_step": 1e3 * strict_elapsed / args.steps}
            out["config"]["native_path_checked"] = bool(native_head and (world > 1 or os.environ.get("ROVMPC_BENCH_TEST_VALIDATE")))
            out["config"]["value_is_fallback"] = not native_head
            out["config"]["native_modes_not_used"] = {str(c): w for c, w in native_why.items()} or None
            out["extra"] = {
                "multi_comm_value": units_per_step * args.steps / steady[3] if 3 in steady else None,
                "single_comm_value": units_per_step * args.steps / steady[1] if 1 in steady else None,
                "note": "steady-state value of the library's own RCCL path with three communicators in rotation / with one; "
                        "`value` is the three-communicator figure when that mode passed its in-run check, else the single-communicator "
                        "one, else (value_is_fallback) torch.distributed's collective"}
        if closed_loop_sharded is not None:
            out["closed_loop"] = closed_loop_sharded
        if kavg_ms:
            achieved = alg_bytes / (kavg_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "rollout_kernel", "kernel_avg_us": kavg_ms * 1e3,
                               "chip_algorithmic_GBps": alg_bytes * args.steps / elapsed / 1e9,
                               "kernel_event_pair_us": kev_ms * 1e3 if kev_ms else None,
                               "kernel_event_pair_min_us": kev_min_ms * 1e3 if kev_min_ms else None,
                               "algorithmic_bytes_per_launch": alg_bytes,
                               "binding_bound": "fp64-valu issue / dependent-chain latency (see valu_f64); HBM is the bound BASELINE.json "
                                                "asks to be reported, not the one that can limit this kernel: at 1 354 FLOP of fp64 per "
                                                "24.4 algorithmic bytes (PMC) even 100 % of the 78.6 TFLOP/s fp64 vector peak moves 78.6e12 / 1354 "
                                                "x 24.4 B = 1.42 TB/s = 18 % of HBM peak, so BASELINE's 40 % HBM target is out of reach by "
                                                "arithmetic; the figures to drive are valu_f64.frac and the chain latency"}
            if (args.N, args.K, args.dtype, world) == (20, 4096, "f64", 1) and args.model == "default" and args.debug_flags == 0:
                bench_extras.attach_pmc(out["roofline"], os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_summary.json"), sha,
                                        kavg_ms * 1e-3, FP64_VECTOR_PEAK_TFLOPS)
            if (args.N, args.K, args.dtype, world) == (50, 16384, "f32", 1) and args.model == "default" and args.debug_flags == 0:
                bench_extras.attach_pmc(out["roofline"], os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_summary_C3.json"), sha,
                                        kavg_ms * 1e-3, FP64_VECTOR_PEAK_TFLOPS, dtype="f32")
        if world == 1 and smpc is None and S == 1 and not args.no_pipelined_extra:
            # extra, not the headline: two independent MPC steps in flight on one GPU (second engine handle
            # on a high-priority stream = its own hardware queue), so one step's launch ramp / arg-min tail
            # overlaps the other's work
            # (256-thread workgroups: four-wave workgroups share a CU three at a time, the single-step geometry's five-wave
            # ones do not, so only then do two launches really run side by side)
            cfg2 = rovmpc.MPCConfig(**{**cfg.__dict__, "threads_per_block": 256 if (args.nt == 0 and args.N * 16 > 256) else args.nt})
            enga = rovmpc.Engine(cfg2, model)
            eng2 = rovmpc.Engine(cfg2, model)
            st2 = torch.cuda.Stream(device=dev, priority=-1)
            pair = [(enga, stream), (eng2, st2)]
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
            eng2.close(); enga.close()
            del st2, pair                     # (every live stream competes for the runtime's few hardware queues)
        default_size = (args.N, args.K, args.dtype, args.model) == (20, 4096, "f64", "default") and args.debug_flags == 0
        if world == 1 and smpc is None and S == 1 and default_size and not args.no_extras:
            bench_extras.run_extras(out, args, cfg, model, dev,
                                    pmc_ctx={"dir": os.path.join(ROOT, "profiles"), "tag": PROFILE_TAG, "sha": sha,
                                             "hbm_peak": HBM_PEAK_GBS, "fp64_peak": FP64_VECTOR_PEAK_TFLOPS})
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.N)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    closed = set()
    for e in list(engines) + [en for en, _ in native.values()] + [eng0]:
        if id(e) not in closed:
            closed.add(id(e))
            e.close()


if __name__ == "__main__":
    main()
