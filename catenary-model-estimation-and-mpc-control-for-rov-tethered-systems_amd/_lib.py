"""ctypes binding of librovmpc.so (include/rovmpc.h).  No CPU fallback: if the HIP library
is missing or no GPU is present every compute entry point raises ``RovmpcError``."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "lib", "librovmpc.so")

F64, F32 = 0, 1
VT_NONE, VT_COMPOSE, VT_TABLE = 0, 1, 2
PREV_INTERP, PREV_HOLD = 0, 1
RK4, EULER, DOUBLE_EULER, TRAPEZOID = 0, 1, 2, 3
ENU, NED = 0, 1
FEATURES_GEN1, FEATURES_GEN2, FEATURES_GEN3 = 0, 1, 2
STATE_LEN = 16

ERR_NAMES = {0: "OK", -1: "ROVMPC_ERR_INVALID", -2: "ROVMPC_ERR_HIP", -3: "ROVMPC_ERR_NO_MODEL",
             -4: "ROVMPC_ERR_UNSUPPORTED"}


class RovmpcError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("dtype", C.c_int32), ("N", C.c_int32),
        ("K", C.c_int32), ("n_shape_pts", C.c_int32), ("vt_mode", C.c_int32), ("prev_mode", C.c_int32),
        ("integrator", C.c_int32), ("frame", C.c_int32), ("force_interpreter", C.c_int32),
        ("candidates_per_block", C.c_int32), ("debug_flags", C.c_int32), ("jit_off", C.c_int32),
        ("feature_map", C.c_int32), ("threads_per_block", C.c_int32), ("no_builtin", C.c_int32),
        ("dt", C.c_double), ("v_scale", C.c_double), ("L", C.c_double), ("cable_wet_weight", C.c_double),
        ("c_lo", C.c_double), ("c_hi", C.c_double),
        ("w_theta", C.c_double), ("w_gamma", C.c_double), ("w_u", C.c_double), ("w_T", C.c_double),
        ("w_taut", C.c_double), ("rho_taut", C.c_double), ("w_floor", C.c_double), ("z_floor", C.c_double),
        ("theta_ref", C.c_double), ("gamma_ref", C.c_double), ("U_ref", C.c_double * 3),
    ]


class State(C.Structure):
    _fields_ = [("P0", C.c_double * 3), ("P1", C.c_double * 3), ("V1", C.c_double * 3), ("A1", C.c_double * 3),
                ("theta", C.c_double), ("gamma", C.c_double), ("theta_prev", C.c_double), ("gamma_prev", C.c_double)]


_P = C.c_void_p
_SIGNATURES = {
    "rovmpc_version": (C.c_char_p, []),
    "rovmpc_default_config": (None, [C.POINTER(Config)]),
    "rovmpc_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "rovmpc_destroy": (None, [_P]),
    "rovmpc_last_error": (C.c_char_p, [_P]),
    "rovmpc_set_model": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32]),
    "rovmpc_set_rotation_table": (C.c_int, [_P, _P]),
    "rovmpc_model_path": (C.c_int32, [_P]),
    "rovmpc_model_structure": (C.c_int32, [_P]),
    "rovmpc_step": (C.c_int, [_P, C.POINTER(State), _P, _P, _P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "rovmpc_rollout_costs": (C.c_int, [_P, C.POINTER(State), _P, _P, _P]),
    "rovmpc_mpc_step_sampled": (C.c_int, [_P, C.POINTER(State), C.c_uint64, C.c_uint64, _P, _P, C.c_int32, _P]),
    "rovmpc_sampled_candidates": (C.c_int, [_P, _P]),
    "rovmpc_sample_candidates_device": (C.c_int, [_P, C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
    "rovmpc_result_len": (C.c_int32, [_P]),
    "rovmpc_step_device": (C.c_int, [_P, _P, _P, _P, _P]),
    "rovmpc_step_device_sharded": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "rovmpc_select_device": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "rovmpc_comm_unique_id": (C.c_int, [_P]),
    "rovmpc_comm_init": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "rovmpc_step_device_allreduce": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P]),
    "rovmpc_comm_join": (C.c_int, [_P, _P]),
    "rovmpc_comm_sync": (C.c_int, [_P, _P]),
    "rovmpc_comm_placement": (C.c_char_p, [_P]),
    "rovmpc_comm_abort": (C.c_int, [_P]),
    "rovmpc_step_batch_device": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    "rovmpc_batch_costs_device": (C.c_int, [_P, C.POINTER(_P)]),
    "rovmpc_set_option": (C.c_int, [_P, C.c_char_p, C.c_double]),
    "rovmpc_device_status": (C.c_int, [_P]),
    "rovmpc_comm_destroy": (C.c_int, [_P]),
    "rovmpc_closed_loop_device": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P, _P]),
    "rovmpc_closed_loop_pipelined_device": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int32, C.c_int32, _P, _P]),
    "rovmpc_timing_enable": (C.c_int, [_P, C.c_int32]),
    "rovmpc_timing_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "rovmpc_predict": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "rovmpc_eval_expression": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int64, _P]),
    "rovmpc_lagrangian_rollout": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, C.c_int64, _P, C.c_int64, _P]),
    "rovmpc_replay": (C.c_int, [_P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_int32, _P, _P]),
    "rovmpc_solve_catenary": (C.c_int, [_P, _P, _P, C.c_double, C.c_int64, _P, _P]),
    "rovmpc_rodrigues": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P]),
    "rovmpc_catenary_points": (C.c_int, [_P, _P, _P, C.c_double, C.c_int64, C.c_int32, _P, _P, _P]),
    "rovmpc_compute_catenary_3d": (C.c_int, [_P, _P, _P, C.c_double, C.c_int64, C.c_int32, _P, _P]),
    "rovmpc_transform_catenary": (C.c_int, [_P, _P, _P, _P, _P, C.c_double, C.c_int64, C.c_int32, _P, _P, _P]),
    "rovmpc_velocity_transform": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "rovmpc_extract_features": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, _P]),
    "rovmpc_gaussian_filter1d": (C.c_int, [_P, _P, C.c_int64, C.c_double, C.c_double, _P]),
    "rovmpc_features_dd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "rovmpc_kabsch_velocity_transform": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen librovmpc.so and bind every symbol of include/rovmpc.h (loud failure if absent)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("ROVMPC_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise RovmpcError(-2, f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 (same SONAME as the system
    # ROCm).  Two HSA runtimes in one process cannot both own the GPU, so when torch is
    # present it is imported first and librovmpc binds to the runtime torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        except AttributeError:
            if path is None and os.environ.get("ROVMPC_LIB_OLD_ABI") == "1":
                continue                 # A/B runs against an older build (tools/ab_bench.sh): entry points it lacks stay unbound
            raise
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def exported_symbols():
    return list(_SIGNATURES)


def check(lib, handle, rc: int):
    if rc != 0:
        msg = lib.rovmpc_last_error(handle)
        raise RovmpcError(rc, msg.decode() if msg else "")
