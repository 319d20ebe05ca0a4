import sys, time, numpy as np
sys.path.insert(0, '.')
import rovmpc
eng = rovmpc.Engine(rovmpc.MPCConfig(N=20, K=4096))
state, U = rovmpc.synthetic_problem(4096, 20)
for _ in range(20): eng.step(state, U)
t0 = time.perf_counter(); n = 300
for _ in range(n): eng.step(state, U)
dt = (time.perf_counter() - t0) / n
print(f"host-pointer rovmpc_step (H2D 1.97 MB + kernel + D2H + sync): {dt*1e6:.1f} us/step = {4096*20/dt/1e9:.3f}e9 horizon-steps/s")
