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
