#!/usr/bin/env python3
"""How the strict bracket (synchronise, K steps, synchronise) scales with K: total = a + b K.  Diagnostic."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rovmpc
cfg = rovmpc.MPCConfig(N=20, K=4096)
dev = torch.device("cuda", 0)
with rovmpc.Engine(cfg) as eng:
    pools = []
    for p in range(4):
        s, U = rovmpc.synthetic_problem(cfg.K, cfg.N, seed=777 + p)
        pools.append(torch.tensor(U, device=dev))
    d_s = torch.tensor(s, device=dev); d_r = torch.empty(eng.result_len, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ps, pr, pu = d_s.data_ptr(), d_r.data_ptr(), [u.data_ptr() for u in pools]
    for i in range(300): eng.step_device(ps, pu[i % 4], pr, stream)
    torch.cuda.synchronize()
    for K in (1, 2, 5, 10, 20, 40, 100, 400):
        ts = []
        for rep in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K): eng.step_device(ps, pu[i % 4], pr, stream)
            t_enq = time.perf_counter()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ts.append((t1 - t0, t_enq - t0))
        tot = np.median([t[0] for t in ts]) * 1e6; enq = np.median([t[1] for t in ts]) * 1e6
        print(f"K={K:4d}: total {tot:8.1f} us = {tot / K:6.2f} us/step; host enqueue loop {enq:8.1f} us ({enq / K:5.2f} us/launch)")
    # the same bracket with HIP events recorded around the K steps (what bench.py's timed region carries)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    def mk(flags):
        e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(e), C.c_uint(flags)) == 0
        return e
    K = 20
    for name, flags in (("none", None), ("torch timing events", "torch"), ("hipEventDefault", 0), ("hipEventDisableSystemFence", 0x20000000),
                        ("hipEventReleaseToDevice", 0x40000000)):
        ts = []
        for rep in range(30):
            if flags == "torch":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            elif flags is not None:
                e0, e1 = mk(flags), mk(flags)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if flags == "torch": e0.record()
            elif flags is not None: hip.hipEventRecord(e0, C.c_void_p(stream))
            for i in range(K): eng.step_device(ps, pu[i % 4], pr, stream)
            if flags == "torch": e1.record()
            elif flags is not None: hip.hipEventRecord(e1, C.c_void_p(stream))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            if flags not in (None, "torch"):
                ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1); hip.hipEventDestroy(e0); hip.hipEventDestroy(e1)
        print(f"K=20 with events [{name}]: {np.median(ts) * 1e6 / K:6.2f} us/step" + (f" (event span {ms.value * 1e3 / K:.2f} us/step)" if flags not in (None, "torch") else ""))
    # bench.py's own sequence: record, K steps, record, spin on the end event's query, synchronise
    for spin in (False, True):
        ts = []
        for rep in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for i in range(K): eng.step_device(ps, pu[i % 4], pr, stream)
            e1.record()
            if spin:
                while not e1.query():
                    pass
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"K=20 torch events, spin on query = {spin}: {np.median(ts) * 1e6 / K:6.2f} us/step (first rep {ts[0] * 1e6 / K:.2f}, min {min(ts) * 1e6 / K:.2f})")
