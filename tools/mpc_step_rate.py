#!/usr/bin/env python3
"""End-to-end rate of the Python surface ``MPC.step(state) -> u`` at the C2 size: candidates sampled on the host
(NumPy, the tensor crosses PCIe every step) vs sampled on the GPU (device_sampling=True)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rovmpc  # noqa: E402

state, _ = rovmpc.synthetic_problem(1, 20)
for dev in (False, True):
    mpc = rovmpc.MPC(N=20, K=4096, device_sampling=dev)
    for _ in range(5):
        mpc.step(state)
    n = 30 if not dev else 500
    t0 = time.perf_counter()
    for _ in range(n):
        mpc.step(state)
    dt = (time.perf_counter() - t0) / n
    print(f"MPC.step, candidates sampled on the {'GPU ' if dev else 'host'}: {dt * 1e6:9.1f} us per step ({1 / dt:8.0f} steps/s)")
    mpc.close()

# the library call alone (no MPC-class bookkeeping): what one control step costs a C caller
import numpy as np  # noqa: E402
mpc = rovmpc.MPC(N=20, K=4096, device_sampling=True)
eng = mpc.engine
m = rovmpc.default_model()
for _ in range(20):
    eng.mpc_step_sampled(state, 1, 0, m.mean[3:6], m.scale[3:6], True)
n = 2000
t0 = time.perf_counter()
for i in range(n):
    eng.mpc_step_sampled(state, 1, i, m.mean[3:6], m.scale[3:6], True)
dt = (time.perf_counter() - t0) / n
print(f"Engine.mpc_step_sampled (one ctypes call, fused sampling): {dt * 1e6:9.1f} us per step")
sp = eng._samp
import ctypes as C  # noqa: E402
fn = eng.lib.rovmpc_mpc_step_sampled
t0 = time.perf_counter()
for i in range(n):
    fn(eng._h, sp["pstate"], 1, i, sp["pm"], sp["ps"], 1, sp["prec"])
dt = (time.perf_counter() - t0) / n
print(f"rovmpc_mpc_step_sampled through ctypes, prepared arguments: {dt * 1e6:9.1f} us per step")
mpc.close()
