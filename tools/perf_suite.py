#!/usr/bin/env python3
"""One process, several configurations of the rollout kernel, HIP-event time per launch -- the A/B instrument of a round.

    python3 tools/perf_suite.py [--only c2,b64,c3,...] [--reps 3] [--tag name]
    ROVMPC_LIB=<pkg>/lib/librovmpc_base.so ROVMPC_LIB_OLD_ABI=1 python3 tools/perf_suite.py      (another build, same box)

Configurations: c2 (N=20 K=4096 f64, the headline), b64 / b8 (batched), c3 (N=50 K=16384 f32), k32k (K=32768 f64),
jit (the reference rows through hiprtc), r59 (rows 5/9), gen2, gen3, c2f32 (C2 size in f32).
Prints one line per configuration (median and minimum over --reps interleaved rounds) and a JSON line at the end.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="c2,b64,c3,k32k,jit,r59")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    import torch
    import rovmpc
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    want = args.only.split(",")
    specs = {
        "c2": dict(N=20, K=4096, dtype="f64", B=1, n=400),
        "c2f32": dict(N=20, K=4096, dtype="f32", B=1, n=400),
        "b8": dict(N=20, K=4096, dtype="f64", B=8, n=100),
        "b64": dict(N=20, K=4096, dtype="f64", B=64, n=30),
        "c3": dict(N=50, K=16384, dtype="f32", B=1, n=200),
        "c3f64": dict(N=50, K=16384, dtype="f64", B=1, n=100),
        "k32k": dict(N=20, K=32768, dtype="f64", B=1, n=100),
        "jit": dict(N=20, K=4096, dtype="f64", B=1, n=300, no_builtin=True),
        "r59": dict(N=20, K=4096, dtype="f64", B=1, n=300, rows=(5, 9)),
        "gen2": dict(N=20, K=4096, dtype="f64", B=1, n=300, gen=2),
        "gen3": dict(N=20, K=4096, dtype="f64", B=1, n=300, gen=3),
    }
    runs = {}
    for name in want:
        sp = specs[name]
        cfg = rovmpc.MPCConfig(N=sp["N"], K=sp["K"], dtype=sp["dtype"])
        model = rovmpc.default_model()
        if sp.get("no_builtin"):
            cfg.no_builtin = True
        if sp.get("rows"):
            model = rovmpc.default_model(*sp["rows"])
        if sp.get("gen") == 2:
            model = rovmpc.generation2_model(); cfg.feature_map = rovmpc.FEATURES_GEN2
        if sp.get("gen") == 3:
            model = rovmpc.generation3_model(); cfg.feature_map = rovmpc.FEATURES_GEN3
        eng = rovmpc.Engine(cfg, model)
        B, N, K = sp["B"], sp["N"], sp["K"]
        tdt = torch.float64 if sp["dtype"] == "f64" else torch.float32
        R = eng.result_len
        npool = 1 if B > 1 else 4
        states = np.empty((max(B, 1), 16)); Us = []
        for p in range(npool):
            U = np.empty((B, K, N, 3), dtype=cfg.np_dtype)
            for b in range(B):
                states[b], U[b] = rovmpc.synthetic_problem(K, N, seed=777 + b + 100 * p, dtype=cfg.np_dtype)
            Us.append(torch.tensor(U, device=dev, dtype=tdt))
        d_states = torch.tensor(states, device=dev)
        d_res = torch.empty((B, R), dtype=torch.float64, device=dev)
        ps, pr, pu = d_states.data_ptr(), d_res.data_ptr(), [u.data_ptr() for u in Us]
        if B > 1:
            fn = lambda i, e=eng, B=B, ps=ps, pu=pu, pr=pr: e.step_batch_device(B, ps, pu[0], pr, stream)
        else:
            fn = lambda i, e=eng, ps=ps, pu=pu, pr=pr, npool=npool: e.step_device(ps, pu[i % npool], pr, stream)
        runs[name] = dict(fn=fn, n=sp["n"], units=B * K * N, eng=eng, keep=(Us, d_states, d_res), times=[])
        for i in range(max(10, sp["n"] // 4)):
            fn(i)
        torch.cuda.synchronize()
    for rep in range(args.reps):
        for name in want:
            r = runs[name]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(5):
                r["fn"](i)
            e0.record()
            for i in range(r["n"]):
                r["fn"](i)
            e1.record()
            torch.cuda.synchronize()
            r["times"].append(e0.elapsed_time(e1) * 1e3 / r["n"])       # us per launch
    out = {"tag": args.tag, "lib": os.environ.get("ROVMPC_LIB", "in-tree")}
    for name in want:
        r = runs[name]
        med, mn = float(np.median(r["times"])), float(np.min(r["times"]))
        out[name] = {"us": med, "us_min": mn, "units_per_s": r["units"] / (med * 1e-6)}
        print(f"{args.tag:>10s} {name:>6s}  {med:9.2f} us (min {mn:9.2f})   {r['units'] / (med * 1e-6):.3e} horizon-steps/s")
        r["eng"].close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
