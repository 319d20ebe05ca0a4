"""rovmpc -- MI355X-native batched MPC rollout engine for tethered-ROV catenary control.

Python call surface of the reference project's hot path (geometry helpers of main_fun.py,
``Catenary(length, reference_frame)``, the open-loop integrators, an ``MPC.step(state) -> u``)
over hand-written HIP kernels behind a C ABI (include/rovmpc.h, lib/librovmpc.so).
There is no CPU fallback: compute entry points raise ``RovmpcError`` without the library/GPU.
"""
from ._lib import (RovmpcError, load_library, exported_symbols, LIB_PATH,
                   F64, F32, VT_NONE, VT_COMPOSE, VT_TABLE, PREV_INTERP, PREV_HOLD, RK4, EULER, DOUBLE_EULER, TRAPEZOID, ENU, NED,
                   FEATURES_GEN1, FEATURES_GEN2, FEATURES_GEN3)
from .expr import compile_expression, disassemble, ExpressionError, Program
from .model import (DynamicsModel, default_model, generation2_model, generation3_model, FEATURE_NAMES_GEN3, load_model_dir, read_equation_csv, select_row,
                    chosen_complexity_from_txt, FEATURE_NAMES_GEN1)
from .engine import Engine, MPCConfig, MPCState, StepResult, default_engine, state_array
from .geometry import (rodrigues_rotation, transform_catenary, transform_catenary_batch, solve_catenary,
                       cable_tension, Catenary, Catenary3D, compute_catenary_3D, lowest_point, rotation_axes, velocity_transform,
                       kabsch_velocity_transform, compute_rotation_kabsch)
from .integrate import (SymbolicRegressor, rk4_integration, integrate_theta_gamma, rk4_theta_gamma,
                        integrate_second_order)
from .features import extract_features, extract_features_arrays, features_dd, features_dd_arrays, preprocess_signals, compute_derivatives
from .lagrangian import euler_lagrange, el_residuals, lagrangian_rollout, differentiate, EulerLagrange
from .trajgen import generate_rov_trajectories, trajectory_csv
from .mpc import MPC, GaussianSampler, DeviceGaussianSampler, synthetic_problem

__version__ = "0.1.0"
