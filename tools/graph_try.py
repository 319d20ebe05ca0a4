#!/usr/bin/env python3
"""K steps submitted as ONE hipGraph (captured from the library's own launches) against K plain launches.  Diagnostic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rovmpc
cfg = rovmpc.MPCConfig(N=20, K=4096); dev = torch.device("cuda", 0)
with rovmpc.Engine(cfg) as eng:
    pools = []
    for p in range(8):
        s, U = rovmpc.synthetic_problem(cfg.K, cfg.N, seed=777 + p); pools.append(torch.tensor(U, device=dev))
    d_s = torch.tensor(s, device=dev); d_r = torch.empty((2, eng.result_len), dtype=torch.float64, device=dev)
    ps, pu = d_s.data_ptr(), [u.data_ptr() for u in pools]
    cur = torch.cuda.current_stream()
    for i in range(300): eng.step_device(ps, pu[i % 8], d_r[i % 2].data_ptr(), cur.cuda_stream)
    torch.cuda.synchronize()
    K = 20
    ref = None
    for mode in ("plain", "graph", "plain", "graph"):
        ts = []
        for rep in range(20):
            g = None
            if mode == "graph":
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    st = torch.cuda.current_stream().cuda_stream
                    for i in range(K): eng.step_device(ps, pu[i % 8], d_r[i % 2].data_ptr(), st)
            torch.cuda.synchronize()
            e1 = torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            if g is not None: g.replay()
            else:
                for i in range(K): eng.step_device(ps, pu[i % 8], d_r[i % 2].data_ptr(), cur.cuda_stream)
            e1.record()
            while not e1.query(): pass
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            out = d_r.cpu().numpy().copy()
            if ref is None: ref = out
            assert np.array_equal(out, ref), "records differ"
        print(f"{mode:6s}: {np.median(ts) * 1e6 / K:6.2f} us/step (min {min(ts) * 1e6 / K:.2f}); records equal to the first run's")
