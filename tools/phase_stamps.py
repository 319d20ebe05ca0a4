#!/usr/bin/env python3
"""Diagnostic: where one rollout-kernel launch spends its time, from in-kernel 100 MHz stamps.
Needs the diagnostic library (make -C <pkg> diag); run with ROVMPC_LIB=<pkg>/lib/librovmpc_diag.so."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rovmpc  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
DBG = int(sys.argv[3]) if len(sys.argv) > 3 else 0
CKB = int(sys.argv[4]) if len(sys.argv) > 4 else 0
eng = rovmpc.Engine(rovmpc.MPCConfig(N=N, K=K, debug_flags=DBG, candidates_per_block=CKB))
state, U = rovmpc.synthetic_problem(K, N)
for _ in range(5):
    eng.step(state, U)
lib = eng.lib
lib.rovmpc_diag_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
buf = np.zeros((K, 8), dtype=np.uint64)
nb = C.c_int32()
assert lib.rovmpc_diag_read_stamps(eng._h, buf.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0
st = buf[:nb.value].astype(np.int64)
t0 = st[:, 0].min()
names = ["start", "U in LDS", "prefix done", "features done", "integration done", "geometry done", "outputs stored", "ticket drawn"]
print(f"{nb.value} workgroups; times in us from the first workgroup's start (100 MHz clock)")
for i, n in enumerate(names):
    col = st[:, i]
    col = col[col > 0]
    print(f"  {n:18s} median {np.median(col - t0) / 100:7.2f}   min {(col.min() - t0) / 100:7.2f}   max {(col.max() - t0) / 100:7.2f}")
