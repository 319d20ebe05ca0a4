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
NTB = int(sys.argv[5]) if len(sys.argv) > 5 else 0
MODEL = sys.argv[6] if len(sys.argv) > 6 else "default"        # default | gen2 | gen3 | rows:CT,CG | jit-default
DTYPE = os.environ.get("STAMPS_DTYPE", "f64")
cfg = rovmpc.MPCConfig(N=N, K=K, debug_flags=DBG, candidates_per_block=CKB, threads_per_block=NTB, dtype=DTYPE)
model = rovmpc.default_model()
if MODEL == "gen2":
    model = rovmpc.generation2_model(); cfg.feature_map = rovmpc.FEATURES_GEN2
elif MODEL == "gen3":
    model = rovmpc.generation3_model(); cfg.feature_map = rovmpc.FEATURES_GEN3
elif MODEL.startswith("rows:"):
    model = rovmpc.default_model(*(int(v) for v in MODEL[5:].split(",")))
elif MODEL == "jit-default":
    cfg.no_builtin = True
eng = rovmpc.Engine(cfg, model)
print("model", MODEL, "->", eng.model_path)
state, U = rovmpc.synthetic_problem(K, N, dtype=cfg.np_dtype)
for _ in range(5):
    eng.step(state, U)
if os.environ.get("STAMPS_BACK_TO_BACK"):        # steady state: launches back to back on one stream, stamps of the last one
    import torch
    dev = torch.device("cuda", 0)
    d_state = torch.tensor(state, device=dev); d_U = torch.tensor(U, device=dev, dtype=torch.float64 if DTYPE == "f64" else torch.float32)
    d_res = torch.empty(eng.result_len, dtype=torch.float64, device=dev)
    for _ in range(int(os.environ["STAMPS_BACK_TO_BACK"])):
        eng.step_device(d_state.data_ptr(), d_U.data_ptr(), d_res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
lib = eng.lib
lib.rovmpc_diag_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
buf = np.zeros((K, 16), dtype=np.uint64)
nb = C.c_int32()
if not os.environ.get("STAMPS_BACK_TO_BACK"):     # clear what the warm-up launches left, then one launch to read
    lib.rovmpc_diag_read_stamps(eng._h, buf.ctypes.data_as(C.c_void_p), C.byref(nb))
    eng.step(state, U)
assert lib.rovmpc_diag_read_stamps(eng._h, buf.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0
st = buf[:nb.value].astype(np.int64)
t0 = st[:, 0].min()
names = ["start", "U in LDS", None, "features done", "integration done", "geometry done", "outputs stored", "ticket drawn",
         "gamma wave: chain starts", "gamma wave: chain done", "gamma wave: sines done", None, "phase 2a done (wave 0)", "phase 2a barrier passed", "phase 5: block arg-min done", "phase 5: barrier passed"]
print(f"{nb.value} workgroups; times in us from the first workgroup's start (100 MHz clock)")
for i, n in enumerate(names):
    if n is None:
        continue
    col = st[:, i]
    col = col[col > 0]
    if len(col) == 0:
        continue
    print(f"  {n:18s} median {np.median(col - t0) / 100:7.2f}   min {(col.min() - t0) / 100:7.2f}   max {(col.max() - t0) / 100:7.2f}")

# slot 11: nibble w = 8 | SIMD of wave w
import collections
pat = collections.Counter()
for v in buf[:nb.value, 11]:
    v = int(v)
    pat["".join(str((v >> (4 * w)) & 3) if (v >> (4 * w)) & 8 else "-" for w in range(8))] += 1
print("SIMD of waves 0..7 (pattern: workgroups):", dict(pat.most_common(8)))
print("per workgroup, us after ITS OWN start (median / p10 / p90):")
for i, n in enumerate(names):
    if n is None or i == 0:
        continue
    m = st[:, i] > 0
    if not m.any():
        continue
    dl = (st[m, i] - st[m, 0]) / 100
    print(f"  {n:32s} {np.median(dl):7.2f} {np.percentile(dl, 10):7.2f} {np.percentile(dl, 90):7.2f}")

# residency: which CU ran each workgroup (HW_ID / XCC_ID stamp), how many were alive together
hw = buf[:nb.value, 2]
cu = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64) * 4096 + ((hw >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64)
start, end = st[:, 0], st[:, 7]
dur = (end - start) / 100
print(f"workgroup lifetime us: median {np.median(dur):.2f}  p10 {np.percentile(dur, 10):.2f}  p90 {np.percentile(dur, 90):.2f}")
ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
print(f"distinct CUs {len(np.unique(cu))}; peak workgroups alive on the chip {np.cumsum(ev[:, 1]).max()}")
peaks = []
for c in np.unique(cu):
    m = cu == c
    e = np.concatenate([np.stack([start[m], np.ones(m.sum(), np.int64)], 1), np.stack([end[m], -np.ones(m.sum(), np.int64)], 1)])
    e = e[np.argsort(e[:, 0], kind="stable")]
    peaks.append(np.cumsum(e[:, 1]).max())
print(f"peak workgroups alive per CU: median {np.median(peaks):.0f}  max {max(peaks)}; workgroups per CU median {np.median(np.bincount(np.unique(cu, return_inverse=True)[1])):.0f}")

# dispatch ramp: start of the workgroups by XCC (hwreg XCC_ID) and by position in the grid
xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
print("start by XCC (us after the first workgroup): " + "  ".join(f"xcc{x}: {np.median(start[xcc == x] - t0) / 100:.2f} [{(start[xcc == x].min() - t0) / 100:.2f}..{(start[xcc == x].max() - t0) / 100:.2f}] n={int((xcc == x).sum())}" for x in np.unique(xcc)))
order = np.argsort(start, kind="stable")
print("start of the k-th workgroup to start: " + "  ".join(f"{k}: {(start[order[k]] - t0) / 100:.2f}" for k in (0, 1, 2, 4, 8, 16, 32, 64, 128, 192, 255) if k < len(order)))
print("start by blockIdx: " + "  ".join(f"{b}: {(start[b] - t0) / 100:.2f}" for b in (0, 1, 2, 7, 8, 9, 16, 64, 128, 255) if b < len(start)))
