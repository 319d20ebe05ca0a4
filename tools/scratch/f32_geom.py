import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, rovmpc, torch
K, N = 1024, 20
state, U = rovmpc.synthetic_problem(K, N, seed=901, dtype=np.float32)
state[12:14] += 0.01
ref = None
for ck, nt in [(0, 0), (16, 320), (16, 256), (16, 384), (16, 512), (32, 512), (64, 512), (32, 256), (8, 192)]:
    with rovmpc.Engine(rovmpc.MPCConfig(N=N, K=K, dtype="f32", candidates_per_block=ck, threads_per_block=nt)) as e:
        J, traj = e.rollout_costs(state, U, return_traj=True)
    if ref is None: ref, tref = J, traj
    d = np.flatnonzero(J != ref)
    dt = np.argwhere(traj != tref)
    print((ck, nt), "J differs in", d.size, d[:6], "traj differs in", len(dt), dt[:3].tolist())
