#!/usr/bin/env python3
"""What an idle torch.cuda.synchronize() costs as streams and engines are added to the process.  Diagnostic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
dev = torch.device("cuda", 0)
x = torch.zeros(16, device=dev)
def idle_sync(label):
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    # and right after a finished kernel (event observed complete by spinning)
    tk = []
    for _ in range(50):
        e = torch.cuda.Event(); x.add_(1); e.record()
        while not e.query(): pass
        t0 = time.perf_counter(); torch.cuda.synchronize(); tk.append(time.perf_counter() - t0)
    print(f"{label:50s} idle sync {np.median(ts) * 1e6:6.1f} us; after a completed kernel {np.median(tk) * 1e6:6.1f} us")
idle_sync("torch only")
s1 = torch.cuda.Stream(); idle_sync("+ one more torch stream (never used)")
with torch.cuda.stream(s1): x.add_(1)
idle_sync("+ that stream used once")
import rovmpc
eng = rovmpc.Engine(rovmpc.MPCConfig(N=20, K=4096))
idle_sync("+ rovmpc engine (its stream + the pipe stream)")
state, U = rovmpc.synthetic_problem(4096, 20)
d_s = torch.tensor(state, device=dev); d_U = torch.tensor(U, device=dev); d_r = torch.empty(eng.result_len, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(50): eng.step_device(d_s.data_ptr(), d_U.data_ptr(), d_r.data_ptr(), st)
idle_sync("+ 50 steps on the current stream")
tk = []
for _ in range(50):
    e = torch.cuda.Event(enable_timing=True)
    for i in range(20): eng.step_device(d_s.data_ptr(), d_U.data_ptr(), d_r.data_ptr(), st)
    e.record()
    while not e.query(): pass
    t0 = time.perf_counter(); torch.cuda.synchronize(); tk.append(time.perf_counter() - t0)
print(f"after 20 steps + timing event observed complete: sync {np.median(tk) * 1e6:6.1f} us")
cur = torch.cuda.current_stream()
for label, waiter in (("event.query spin", lambda e: e.query()), ("stream.query spin", lambda e: cur.query()), ("both", lambda e: (cur.query(), e.query())[1])):
    tk, tt = [], []
    for _ in range(50):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        t00 = time.perf_counter()
        e0.record()
        for i in range(20): eng.step_device(d_s.data_ptr(), d_U.data_ptr(), d_r.data_ptr(), st)
        e.record()
        while not waiter(e): pass
        t0 = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter(); tk.append(t1 - t0); tt.append(t1 - t00)
    print(f"20 steps, wait by {label:18s}: final sync {np.median(tk) * 1e6:6.1f} us, whole bracket {np.median(tt) * 1e6 / 20:6.2f} us/step")
for label in ("device sync only", "stream.synchronize then device sync", "event.synchronize then device sync"):
    tk, tt = [], []
    for _ in range(50):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        t00 = time.perf_counter()
        e0.record()
        for i in range(20): eng.step_device(d_s.data_ptr(), d_U.data_ptr(), d_r.data_ptr(), st)
        e.record()
        while not e.query(): pass
        t0 = time.perf_counter()
        if label.startswith("stream"): cur.synchronize()
        if label.startswith("event"): e.synchronize()
        torch.cuda.synchronize(); t1 = time.perf_counter(); tk.append(t1 - t0); tt.append(t1 - t00)
    print(f"20 steps, {label:38s}: tail {np.median(tk) * 1e6:6.1f} us, whole bracket {np.median(tt) * 1e6 / 20:6.2f} us/step")
