#!/usr/bin/env python3
"""Launch-geometry sweep of the fused rollout kernel: candidates per workgroup x threads per workgroup,
step time from HIP events on the launch stream.  Results do not depend on the geometry (tests cover that);
this is where the library's automatic choice (configure_geometry in csrc/rovmpc.hip) comes from.
  python tools/geometry_sweep.py N K dtype [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import rovmpc  # noqa: E402

N = int(sys.argv[1]); K = int(sys.argv[2]); dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
tdt = torch.float64 if dtype == "f64" else torch.float32
state, U = rovmpc.synthetic_problem(K, N)
d_state = torch.tensor(state, dtype=torch.float64, device="cuda")
d_U = torch.tensor(U, dtype=tdt, device="cuda")
s = torch.cuda.current_stream().cuda_stream
rows = []
for ck in (0, 8, 16, 32, 64):
    for nt in (0, 128, 192, 256, 320, 384, 512):
        if (ck == 0) != (nt == 0):
            continue
        try:
            eng = rovmpc.Engine(rovmpc.MPCConfig(N=N, K=K, dtype=dtype,
                                                 candidates_per_block=ck, threads_per_block=nt))
        except rovmpc.RovmpcError as e:
            rows.append({"ck": ck, "nt": nt, "error": str(e)[:60]}); continue
        d_out = torch.empty(eng.result_len, dtype=torch.float64, device="cuda")
        for _ in range(5):
            eng.step_device(d_state.data_ptr(), d_U.data_ptr(), d_out.data_ptr(), s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            eng.step_device(d_state.data_ptr(), d_U.data_ptr(), d_out.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / steps
        rows.append({"ck": ck or "auto", "nt": nt or "auto", "us": round(us, 2), "units_per_s": round(K * N / us * 1e6)})
        print(rows[-1], flush=True)
        del eng
best = min((r for r in rows if "us" in r), key=lambda r: r["us"])
print(json.dumps({"N": N, "K": K, "dtype": dtype, "best": best}))
