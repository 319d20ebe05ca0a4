/*
 * rovmpc.h -- C ABI of librovmpc.so: the MI355X (gfx950) batched MPC rollout engine for
 * the tether (theta, gamma) state of a tethered ROV.
 *
 * This is the drop-in boundary for ONE hot path of the reference project
 * (eather0056/Catenary-Model-Estimation-and-MPC-Control-for-ROV-Tethered-Systems).  The
 * reference is pure Python; every entry point below names the Python callable (file:line,
 * relative to the reference root) whose per-row loop it replaces.  The binding a
 * maintainer of the reference would add is a ctypes stub -- see INTEGRATION.md.
 *
 * Conventions
 *  - every function returns ROVMPC_OK (0) or a negative ROVMPC_ERR_* code and never throws;
 *    rovmpc_last_error() gives the message of the last failure on that handle
 *    (or of the last failed rovmpc_create when passed NULL);
 *  - all buffers are caller-owned and C-contiguous.  Functions without a `_device` suffix
 *    take HOST pointers and stage H2D/D2H themselves (blocking).  `_device` functions take
 *    DEVICE pointers plus a hipStream_t (as void*), enqueue asynchronously and do not
 *    synchronise;
 *  - "real" buffers (U, J, traj_all) are double when cfg.dtype == ROVMPC_F64 and float when
 *    ROVMPC_F32; state, results and the geometry helpers are always double;
 *  - one handle = one device + one internal stream + one workspace; a handle is NOT
 *    thread-safe, distinct handles are independent;
 *  - numeric failure is reported in-band as NaN exactly where the reference yields NaN
 *    (solve_catenary: no sign change on the bracket).
 */
#ifndef ROVMPC_H
#define ROVMPC_H

#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#else   /* hiprtc has no <stdint.h>: the compiler's own type macros give the same typedefs */
typedef __INT32_TYPE__ int32_t;
typedef __UINT32_TYPE__ uint32_t;
typedef __INT64_TYPE__ int64_t;
typedef __UINT64_TYPE__ uint64_t;
typedef __UINTPTR_TYPE__ uintptr_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define ROVMPC_OK               0
#define ROVMPC_ERR_INVALID     -1   /* bad argument / configuration                        */
#define ROVMPC_ERR_HIP         -2   /* a HIP runtime call failed (no GPU, OOM, ...)        */
#define ROVMPC_ERR_NO_MODEL    -3   /* rovmpc_set_model has not been called                */
#define ROVMPC_ERR_UNSUPPORTED -4   /* valid request this build cannot serve               */

#define ROVMPC_F64 0
#define ROVMPC_F32 1

#define ROVMPC_VT_NONE    0   /* candidate controls already are rob_cor_speed              */
#define ROVMPC_VT_COMPOSE 1   /* v_cat = R_theta(+theta) R_gamma(-gamma) v_world per step  */
#define ROVMPC_VT_TABLE   2   /* v_cat = R[n] v_world, R supplied per horizon step         */

#define ROVMPC_PREV_INTERP 0  /* delay slots x16,x17 follow the reference's row midpoint   */
#define ROVMPC_PREV_HOLD   1  /* delay slots held at the previous node for all RK4 stages  */

#define ROVMPC_RK4   0        /* simulate_rk4_theta_gamma.py:52-68                         */
#define ROVMPC_EULER 1        /* main_fun.py:735-764                                       */
#define ROVMPC_DOUBLE_EULER 2 /* replay only: second-derivative models, test_cluster.py:110-129 */
#define ROVMPC_TRAPEZOID 3    /* replay only: cumulative_trapezoid + cumsum, dd_cluster.py:221-226 */

#define ROVMPC_ENU 0
#define ROVMPC_NED 1

/* Feature map the loaded expressions are written over:
 * GEN1: 18 slots [P1, V1, A1, unit_rel, tension, angle_proj, theta, gamma, theta_prev, gamma_prev],
 *       StandardScaler-normalised (simply.py:15-41, saved_models/);
 * GEN2: 17 slots [P1, V1, A1, unit_rel, theta, gamma, cos(theta), sin(gamma), angle_proj(unclipped)],
 *       unscaled (simulate_rk4_theta_gamma.py:12-42, outputs/differential_training_new_feature/);
 * GEN3: 14 slots [theta, gamma, dtheta, dgamma, v_sway, v_surge, a_sway, a_surge, V(3), a(3)] (features_dd,
 *       main_fun.py:811-871; V in m/s), StandardScaler-normalised; the expressions are SECOND derivatives
 *       (ddtheta, ddgamma -- dd_cluster.py, outputs/dd_C6_all_50_s_20250511_013928/).  The rollout integrates
 *       y = (theta, gamma, dtheta, dgamma), y' = (dtheta, dgamma, f_theta, f_gamma) with RK4, or with the
 *       reference's explicit double Euler (test_cluster.py:110-129) when integrator = ROVMPC_EULER;
 *       rovmpc_state.theta_prev / gamma_prev carry dtheta / dgamma at node 0 in this map. */
#define ROVMPC_FEATURES_GEN1 0
#define ROVMPC_FEATURES_GEN2 1
#define ROVMPC_FEATURES_GEN3 2

#define ROVMPC_MAX_FEATURES 32
#define ROVMPC_MAX_CODE     256
#define ROVMPC_MAX_STACK    16
#define ROVMPC_STATE_LEN    16

/* Bytecode of one symbolic expression: postfix, one int32 per instruction,
 * (arg << 8) | opcode.  arg = feature index (PUSH_F), constant index (PUSH_C) or integer
 * exponent (POWI).  Operator vocabulary = every operator the reference's PySR runs use
 * (simply.py:65-66, PySRTrainingScript.py:53-54, cluster_run/train_dynamics.py:28-46,
 * dynamic_eq_theta_cluster.py:35-43). */
enum rovmpc_opcode {
    ROVMPC_OP_PUSH_C = 0, ROVMPC_OP_PUSH_F = 1,
    ROVMPC_OP_ADD = 2, ROVMPC_OP_SUB = 3, ROVMPC_OP_MUL = 4, ROVMPC_OP_DIV = 5,
    ROVMPC_OP_NEG = 6, ROVMPC_OP_SIN = 7, ROVMPC_OP_COS = 8, ROVMPC_OP_TANH = 9,
    ROVMPC_OP_ABS = 10, ROVMPC_OP_SQUARE = 11, ROVMPC_OP_EXP = 12, ROVMPC_OP_LOG = 13,
    ROVMPC_OP_SQRT = 14, ROVMPC_OP_POW = 15, ROVMPC_OP_POWI = 16,
    ROVMPC_OP_SAFE_LOG = 17,   /* log(|x| + 1e-5)  */
    ROVMPC_OP_SAFE_SQRT = 18,  /* sqrt(|x|)        */
    ROVMPC_OP_COUNT = 19
};

typedef struct rovmpc_config {
    int32_t struct_size;        /* = sizeof(rovmpc_config), ABI check                       */
    int32_t device;             /* HIP device ordinal                                       */
    int32_t dtype;              /* ROVMPC_F64 | ROVMPC_F32                                  */
    int32_t N;                  /* horizon steps                                            */
    int32_t K;                  /* candidates per step on this handle (this shard)          */
    int32_t n_shape_pts;        /* M samples of the augmented catenary per node (>= 2)      */
    int32_t vt_mode;            /* ROVMPC_VT_*                                              */
    int32_t prev_mode;          /* ROVMPC_PREV_*                                            */
    int32_t integrator;         /* ROVMPC_RK4 | ROVMPC_EULER                                */
    int32_t frame;              /* ROVMPC_ENU | ROVMPC_NED (catenary.py:10)                 */
    int32_t force_interpreter;  /* 1: never take the compiled-in default-equation path      */
    int32_t candidates_per_block; /* 0 = auto; else a power of two <= 64                    */
    int32_t debug_flags;        /* diagnostics only (phase ablation for profiling); keep 0  */
    int32_t jit_off;            /* 1: never specialise a loaded model with hiprtc           */
    int32_t feature_map;        /* ROVMPC_FEATURES_GEN1 | _GEN2 | _GEN3                     */
    int32_t threads_per_block;  /* 0 = auto; else a multiple of 64 in 64..512                */
    int32_t no_builtin;         /* 1: never substitute the compiled-in kernel, even for the reference's chosen rows
                                   (they then run through hiprtc / the interpreter like any other model)            */
    double dt;                  /* horizon step [s]                                         */
    double v_scale;             /* velocity unit -> m/s (1e-3: mm/s, cf. main_fun.py:815)   */
    double L;                   /* cable length [m] (test_cluster.py:22)                    */
    double cable_wet_weight;    /* [N] (test_cluster.py:23)                                 */
    double c_lo, c_hi;          /* brentq bracket of main_fun.py:425 (1e-6, 10)             */
    double w_theta, w_gamma, w_u, w_T, w_taut, rho_taut, w_floor, z_floor;
    double theta_ref, gamma_ref;
    double U_ref[3];
} rovmpc_config;

/* 16 doubles: P0[3] anchor (rod_end, m), P1[3] ROV attach point (m), V1[3] current
 * rob_cor_speed, A1[3] current acceleration feature, theta, gamma, theta_prev, gamma_prev
 * (feature slots x0..x8, x14..x17 of simply.py:41 at horizon node 0). */
typedef struct rovmpc_state {
    double P0[3], P1[3], V1[3], A1[3];
    double theta, gamma, theta_prev, gamma_prev;
} rovmpc_state;

typedef struct rovmpc_handle rovmpc_handle;

const char *rovmpc_version(void);
void rovmpc_default_config(rovmpc_config *cfg);

int  rovmpc_create(const rovmpc_config *cfg, rovmpc_handle **out);
void rovmpc_destroy(rovmpc_handle *h);
const char *rovmpc_last_error(const rovmpc_handle *h);

/* Learned dynamics = saved_models/scaler.pkl (mean_, scale_) + one row of
 * saved_models/equations_dtheta_dt.csv and one of equations_dgamma_dt.csv compiled to
 * bytecode by the host (replaces PySRRegressor.predict on those rows). */
int rovmpc_set_model(rovmpc_handle *h, int32_t n_features,
                     const double *mean, const double *scale,
                     const int32_t *code_theta, int32_t n_code_theta,
                     const int32_t *code_gamma, int32_t n_code_gamma,
                     const double *consts, int32_t n_consts);

/* ROVMPC_VT_TABLE only: R[N][3][3] row-major, rows [exc1 eyc1 ezc1; exc2 ..; exc3 ..]
 * (batch_correct_velocity.py:38-45). */
int rovmpc_set_rotation_table(rovmpc_handle *h, const double *R);

/* How the loaded model is evaluated inside the rollout kernel: 0 = compiled-in rows of
 * saved_models/eq_*.txt, 1 = bytecode interpreter, 2 = hiprtc-specialised kernel generated from
 * the bytecode at rovmpc_set_model (falls back to 1 if hiprtc is unavailable; the reason is then
 * in rovmpc_last_error). */
int32_t rovmpc_model_path(const rovmpc_handle *h);
/* Structure the code generator found in a hiprtc-specialised model's rows, from their slot-dependency sets (0 otherwise):
 * bit 0: dgamma/dt reads nothing but x15 / x17 (gamma, gamma_prev) -- the gamma path is the same for every candidate and is
 *        integrated once per workgroup on a wave of its own, as for the reference's chosen rows;
 * bit 1: dtheta/dt reads neither x14 nor x15 -- its four RK4 slopes need no stage loop (used together with bit 0).
 * Generation-1 feature map only; environment ROVMPC_JIT_NO_STRUCT=1 switches the analysis off. */
int32_t rovmpc_model_structure(const rovmpc_handle *h);

/* ---- the hot path ---------------------------------------------------------------------
 * One MPC step: roll every candidate control sequence U[K][N][3] over the horizon
 * (closed-loop RK4 of the learned dtheta/dt, dgamma/dt -- simulate_rk4_theta_gamma.py:52-68
 * generalised to feed the state back; catenary parameter/tension -- main_fun.py:418-431,
 * 302-305; theta/gamma-augmented catenary lowest point -- main_fun.py:38-111,
 * fully_augmented_catenary.py:21-22; velocity transform -- velocity_transform_batch.py:
 * 100-101), accumulate the cost, arg-min with np.argmin tie-break.
 * u_out[3] = U[k*][0][:], traj_out[(N+1)][2] = predicted (theta, gamma) of k*. */
int rovmpc_step(rovmpc_handle *h, const rovmpc_state *state, const void *U,
                double *u_out, double *traj_out, double *best_cost, int64_t *best_idx);

/* MPC.step with the proposal drawn on the GPU -- one call per control step, nothing but the 16-double state and the
 * record crosses PCIe: candidates U[k][n][c] = mean[c] + std[c] z, z standard normal from Philox4x32-10 keyed by
 * (seed, step) with Box-Muller (exact law in util_kernels.h; restated in the oracle), candidate 0 = the previous winner
 * shifted by one step when warm_start != 0 (and a previous step exists); then the fused rollout; record_out
 * [result_len] on return.  rovmpc_sampled_candidates copies the tensor of the last such step to the host (tests);
 * rovmpc_sample_candidates_device only fills d_U[K][N][3] on `stream`. */
int rovmpc_mpc_step_sampled(rovmpc_handle *h, const rovmpc_state *state, uint64_t seed, uint64_t step,
                            const double *mean3, const double *std3, int32_t warm_start, double *record_out);
int rovmpc_sampled_candidates(rovmpc_handle *h, void *U_out);
int rovmpc_sample_candidates_device(rovmpc_handle *h, uint64_t seed, uint64_t step, const double *mean3,
                                    const double *std3, void *d_U, void *stream);

/* Parity/debug: all K costs (and, if traj_all != NULL, all K trajectories [K][N+1][2]). */
int rovmpc_rollout_costs(rovmpc_handle *h, const rovmpc_state *state, const void *U,
                         void *J_out, void *traj_all);

/* Result record length in doubles: 5 + 2 (N+1):
 * [J*, k* (as double), u[3], theta_0, gamma_0, ..., theta_N, gamma_N]. */
int32_t rovmpc_result_len(const rovmpc_handle *h);

/* Device-resident, asynchronous flavour: d_state = 16 doubles, d_U = K*N*3 reals,
 * d_result = rovmpc_result_len doubles, all in device memory; stream = hipStream_t. */
int rovmpc_step_device(rovmpc_handle *h, const double *d_state, const void *d_U,
                       double *d_result, void *stream);

/* B independent MPC problems in ONE launch (grid = B x workgroups; every problem has its own sweeper, hand-off granules
 * and record): d_states[B][16], d_U[B][K][N][3], d_results[B][result_len].  The natural multi-ROV / multi-timestep form
 * of the reference's per-frame loop over independent states (catenary_from_data.py:40-50), and the way to fill the
 * chip when one problem (K = 4096) is only one workgroup per CU.  Problem b's record is bit-identical to
 * rovmpc_step_device on (d_states[b], d_U[b]).  rovmpc_batch_costs_device returns the device pointer of the costs
 * J[B][K] of the last batched launch (reals of cfg.dtype; valid until the next launch on the handle). */
int rovmpc_step_batch_device(rovmpc_handle *h, int32_t B, const double *d_states, const void *d_U,
                             double *d_results, void *stream);
int rovmpc_batch_costs_device(rovmpc_handle *h, const void **d_J);

/* Handle options by name: "handoff_timeout_ms" (give-up time of the GPU-side waits of the sharded step, default 10000);
 * test hooks "inject_skip_rolled" / "inject_skip_consumed" = n: the next n sharded steps lose that publication. */
int rovmpc_set_option(rovmpc_handle *h, const char *name, double value);

/* Non-blocking: ROVMPC_ERR_HIP (and the reason in rovmpc_last_error) if a kernel of this handle raised its error word
 * (a hand-off wait that gave up) in work the host has already synchronised with; clears the word. */
int rovmpc_device_status(rovmpc_handle *h);

/* Candidate-sharded step (one handle per rank): writes this rank's record, order-preserving
 * mapped to int64 (k* offset by k_offset), into d_slots[world][result_len] with every
 * other rank's row = INT64_MAX, so that ONE all-reduce(min) over the buffer assembles all
 * ranks' records on every rank.  rovmpc_select_device then takes the lexicographic
 * (cost, index) minimum -- identical to a single np.argmin over all shards. */
int rovmpc_step_device_sharded(rovmpc_handle *h, const double *d_state, const void *d_U,
                               int64_t k_offset, int32_t rank, int32_t world,
                               int64_t *d_slots, void *stream);
int rovmpc_select_device(rovmpc_handle *h, const int64_t *d_slots, int32_t world,
                         double *d_result, void *stream);

/* ---- native collective: the sharded step with RCCL called from the library -----------------
 * librccl is opened with dlopen at rovmpc_comm_init (no link-time dependency).  Rank 0 obtains a
 * 128-byte id (rovmpc_comm_unique_id), the host distributes it to every rank out of band
 * (torch.distributed / MPI / a file), every rank calls rovmpc_comm_init on its handle.
 * rovmpc_step_device_allreduce then enqueues, without synchronising: the fused rollout kernel on
 * `stream`, ONE ncclAllReduce(ncclMin, ncclInt64, world * result_len) and the select kernel on
 * the handle's side streams; d_result is valid once rovmpc_comm_join'ed.  ROVMPC_COMM_SLOTS slot
 * buffers rotate so the collectives of the last steps overlap the next rollouts; the caller
 * must keep at least that many d_result buffers in rotation.  `stream` receives rollout kernels
 * only: the kernel publishes its row with a sequence number that the side stream polls (no event
 * record / cross-stream wait on the caller's timeline).  Up to three communicators (environment
 * ROVMPC_COMMS, default 3) serve the slots in turn so consecutive collectives overlap each other; the streams they use are
 * chosen at the first step by a placement probe against `stream` (rovmpc_comm_placement below), so keep calling with the
 * same stream. */
#define ROVMPC_COMM_SLOTS 4
int rovmpc_comm_unique_id(void *id128);
int rovmpc_comm_init(rovmpc_handle *h, const void *id128, int32_t rank, int32_t world);
int rovmpc_step_device_allreduce(rovmpc_handle *h, const double *d_state, const void *d_U,
                                 int64_t k_offset, double *d_result, void *stream);
/* rovmpc_comm_join: make `stream` wait for the collective streams (does not block the host); returns ROVMPC_ERR_HIP if a
 * hand-off of an already executed step gave up.  rovmpc_comm_sync: join + synchronise `stream` + report such give-ups of
 * every step enqueued so far: a wait that timed out (rollout row never published, slot row never freed) makes that
 * step's record carry a NaN cost, leaves the slot row untouched, and turns this call into an error -- never a silent
 * wrong record. */
int rovmpc_comm_join(rovmpc_handle *h, void *stream);
int rovmpc_comm_sync(rovmpc_handle *h, void *stream);
int rovmpc_comm_destroy(rovmpc_handle *h);
/* From another thread, while steps are in flight: ncclCommAbort on the handle's communicators, so that a collective that
 * will never complete (a dead peer, an ordering fault) lets go of its stream; the hand-off waits behind it time out, the
 * affected records carry NaN costs, rovmpc_comm_sync returns ROVMPC_ERR_HIP.  The handle issues no collective afterwards
 * (rovmpc_comm_destroy, then rovmpc_comm_init or another path). */
int rovmpc_comm_abort(rovmpc_handle *h);
/* Where the collective streams went.  At the first rovmpc_step_device_allreduce the library probes candidate streams
 * (normal and high priority) against `stream` and against each other and keeps, per communicator, one on which a waiting
 * kernel does not delay kernel completion on the others (two hardware queues on one command-processor pipe do that to
 * each other: rollouts of 55 us instead of 20, measured).  Returns a one-line report of the probe ("" before it ran);
 * environment ROVMPC_COMM_PLACE=0 switches the probe off. */
const char *rovmpc_comm_placement(const rovmpc_handle *h);

/* Closed loop, device resident (BASELINE config 5): for i = 0..T-1 enqueue, without any host
 * synchronisation, (1) the plant update -- state = exo[i] (16 doubles per step in rovmpc_state
 * order: the measured P0, P1, V1, A1, theta, gamma, theta_prev, gamma_prev); with feedback != 0,
 * from the second step on, theta/gamma_prev = theta/gamma and (theta, gamma) = the first predicted
 * node of the previous step's winner instead of the measured values -- and (2) one MPC step on
 * candidate batch pools[i % n_pools], record into results[i][result_len].  If rovmpc_comm_init
 * was called the step is the sharded one (all-reduce(min) per step; with feedback the plant
 * update then waits for the global record). */
int rovmpc_closed_loop_device(rovmpc_handle *h, const double *d_exo, int64_t T, double *d_state,
                              const void *d_pools, int32_t n_pools, int64_t k_offset,
                              int32_t feedback, double *d_results, void *stream);

/* The same loop (single GPU, no communicator), pipelined: one step per launch, launches alternating between two internal streams (forked from and joined
 * to `stream`).  Launch i + 1 starts while launch i runs -- its launch latency, dispatch ramp and the previous sweeper's
 * epilogue leave the critical path -- and its workgroups wait, on the GPU, for the state launch i's sweeper publishes.
 * Records are bit-identical to rovmpc_closed_loop_device's.  ROVMPC_ERR_UNSUPPORTED when two grids do not fit the chip
 * at once, with a communicator, or for the bytecode interpreter. */
int rovmpc_closed_loop_pipelined_device(rovmpc_handle *h, const double *d_exo, int64_t T, double *d_state,
                                        const void *d_pools, int32_t n_pools, int32_t feedback, double *d_results,
                                        void *stream);

/* Per-launch timing of the rollout kernel with HIP events on the launch stream. */
int rovmpc_timing_enable(rovmpc_handle *h, int32_t max_launches);
int rovmpc_timing_read(rovmpc_handle *h, double *avg_ms, double *min_ms, int32_t *count);

/* ---- batched mirrors of the reference's per-row helpers (host pointers) ---------------- */

/* model.predict(X(n,F)) for the loaded theta (which=0) / gamma (which=1) expression;
 * X is what the reference feeds its models: already-scaled feature rows. */
int rovmpc_predict(rovmpc_handle *h, const double *Xs, int64_t n, int32_t which, double *out);

/* Any validated bytecode program on n rows X[n][F] (what sympy.lambdify'd expressions are used for in the reference's
 * evaluation scripts); in particular the Euler-Lagrange residuals EOM_theta / EOM_gamma of a discovered Lagrangian on the
 * rows (theta, gamma, dtheta, dgamma, ddtheta, ddgamma) of a trajectory -- lagrangian_pipeline_old.py:60-90,
 * LagrangianModelEstimator.py:158-195; the host derives the two EOM expressions (lagrangian.py). */
int rovmpc_eval_expression(rovmpc_handle *h, const int32_t *code, int32_t n_code, const double *consts, int32_t n_consts,
                           const double *X, int32_t F, int64_t n, double *out);

/* evaluate_lagrangian_on_test.py:59-68: forward integration of the accelerations solved from the Euler-Lagrange equations,
 * programs over x0..x3 = (theta, gamma, vtheta, vgamma); B rollouts y0[B][4] over one time grid; out[4][B][T] =
 * theta, gamma, vtheta, vgamma. */
int rovmpc_lagrangian_rollout(rovmpc_handle *h, const int32_t *code_theta, int32_t n_code_theta, const int32_t *code_gamma,
                              int32_t n_code_gamma, const double *consts, int32_t n_consts, const double *time, int64_t T,
                              const double *y0, int64_t B, double *out);

/* rk4_integration(model, x_input, time, y0) (simulate_rk4_theta_gamma.py:52-68) when
 * integrator == ROVMPC_RK4, integrate_theta_gamma (main_fun.py:735-764) when ROVMPC_EULER;
 * both expressions in one call, either output may be NULL.  Xs[T][F] scaled rows.
 * ROVMPC_DOUBLE_EULER / ROVMPC_TRAPEZOID treat the expressions as second derivatives
 * (theta'', gamma'') and integrate twice from zero angular velocity, as the reference's
 * second-order evaluation scripts do (test_cluster.py:110-129, dd_cluster.py:221-226). */
int rovmpc_replay(rovmpc_handle *h, const double *Xs, const double *time, int64_t T,
                  double theta0, double gamma0, int32_t integrator,
                  double *theta_out, double *gamma_out);

/* solve_catenary(l, delta_H, L) (main_fun.py:418-431) and, if T_out != NULL, the tension
 * rule of main_fun.py:302-305 with cfg.cable_wet_weight. */
int rovmpc_solve_catenary(rovmpc_handle *h, const double *l, const double *delta_H, double L,
                          int64_t n, double *C_out, double *T_out);

/* rodrigues_rotation(vector, axis, angle_rad) (main_fun.py:18-35), n rows. */
int rovmpc_rodrigues(rovmpc_handle *h, const double *v, const double *axis,
                     const double *angle, int64_t n, double *out);

/* Catenary(length=L, reference_frame=cfg.frame)(a, b) (catenary.py:10,25-29) for n pairs:
 * pts[n][M][3]; valid[n] = 0 where the reference's catenary_fn would return None at [3]
 * (then pts row = NaN); params[n][3] = (C, sag, x_low). */
int rovmpc_catenary_points(rovmpc_handle *h, const double *A, const double *B, double L,
                           int64_t n, int32_t M, double *pts, int32_t *valid, double *params);

/* compute_catenary_3D(p0, p1, rope_length, num_points) (models/catenary_3d.py:5-39), the catenary generator the reference
 * itself holds, for n pairs: pts[n][num_points][3] from p0 to p1 (the straight np.linspace when rope_length <= |p1 - p0|,
 * :13-14); a_out[n] (may be NULL) = the catenary parameter the fixed point of :18-24 stopped at, NaN for the straight case.
 * Usable as the `catenary_fn` of transform_catenary (the host layer's Catenary3D), which pins the whole augmented-catenary
 * path to reference code. */
int rovmpc_compute_catenary_3d(rovmpc_handle *h, const double *p0, const double *p1, double rope_length,
                               int64_t n, int32_t num_points, double *pts, double *a_out);

/* transform_catenary(point_A, point_B, Catenary(L), theta, gamma) (main_fun.py:38-111):
 * out[4][n][M][3] = original, theta_rotated, theta_aligned, final; npts[n][2] = number of
 * meaningful rows in out[0] and out[1..3] (M, or 2 for the straight-segment fallback of
 * main_fun.py:67-69; remaining rows NaN); z_low[n] = lowest z of `final`
 * (fully_augmented_catenary.py:21-22). */
int rovmpc_transform_catenary(rovmpc_handle *h, const double *A, const double *B,
                              const double *theta, const double *gamma, double L,
                              int64_t n, int32_t M, double *out, int32_t *npts, double *z_low);

/* R @ v per row (velocity_transform_batch.py:100-101): R[n][3][3], v[n][3]. */
int rovmpc_velocity_transform(rovmpc_handle *h, const double *R, const double *v,
                              int64_t n, double *out);

/* features_dd (main_fun.py:811-871), the feature / target table of the second-order runs, for T rows of a log:
 * P0_mm, P1_mm, V_mm [T][3] as logged (mm, mm/s: the map divides by 1000), time [T], raw theta, gamma [T] ->
 * features[T][14] = [savgol(theta), savgol(gamma), their np.gradient, v_sway, v_surge, np.gradient of both,
 * V (m/s), np.gradient(V)] and targets[T][2] = second np.gradient of the smoothed angles (targets may be null).
 * Savitzky-Golay as scipy.signal.savgol_filter(x, window, polyorder) with mode='interp' (reference: 11, 3);
 * T >= window. */
/* extract_features (simply.py:15-41; main_fun.py:167-193 when with_prev = 0) for T rows:
 * P0, P1 [T][3] in metres, V1 [T][3], time [T], theta [T], gamma [T] ->
 * out[T][18] (or [T][16]) = [P1, V1, A1 = np.gradient(V1, time), unit_rel, tension, angle_proj,
 * theta, gamma (, theta_prev, gamma_prev)]; np.gradient's second-order non-uniform interior
 * stencil and first-order edges. */
int rovmpc_features_dd(rovmpc_handle *h, const double *P0_mm, const double *P1_mm, const double *V_mm,
                       const double *time, const double *theta, const double *gamma, int64_t T,
                       int32_t window, int32_t polyorder, double *features, double *targets);

/* scipy.ndimage.gaussian_filter1d(x, sigma, truncate=truncate) with its default 'reflect' boundary
 * (preprocess_signals, main_fun.py:768-776: sigma 2, truncate 4; build_theta_features_valid :510-518). */
int rovmpc_gaussian_filter1d(rovmpc_handle *h, const double *x, int64_t T, double sigma, double truncate, double *out);

int rovmpc_extract_features(rovmpc_handle *h, const double *P0, const double *P1, const double *V1,
                            const double *time, const double *theta, const double *gamma,
                            int64_t T, int32_t with_prev, double *out);

/* velocity_transform_batch.py:8-19,71-107 (batch_gates = 1) / velocity_transform.py:42-80
 * (batch_gates = 0) for T frames of M cable markers: R = Kabsch rotation of the centred marker
 * sets P[T][M][3] -> Q[T][M][3] (3x3 SVD with the reflection fix), v_out = R @ v.
 * Frames with a non-finite marker, M < 3 or (batch_gates) |P - Q|_F < 1e-6 yield NaN rows,
 * as the reference writes.  R_out[T][9] may be NULL. */
int rovmpc_kabsch_velocity_transform(rovmpc_handle *h, const double *P, const double *Q, const double *v,
                                     int64_t T, int32_t M, int32_t batch_gates, double *v_out, double *R_out);

#ifdef __cplusplus
}
#endif
#endif /* ROVMPC_H */
