// rollout_kernels.h -- the fused MPC rollout kernel and its arg-min epilogue (gfx950).
//
// One workgroup owns CK consecutive candidates and ALL N horizon steps of them.  The
// candidate-control chunk U[k0:k0+CK][N][3] is contiguous in HBM and is read exactly once,
// coalesced, into LDS; everything else lives in LDS/registers until the K costs, the
// per-block best and (optionally) the trajectories are written.
//
//   phase 0  U chunk -> LDS (coalesced)                                  all threads
//   phase 2  node positions P_n = P_0 + sum_{j<n} (v_scale dt) U_j,      (N+1)*CK items
//            exogenous feature rows (simply.py:25-31), scaled (scaler.pkl),
//            rotation axes (main_fun.py:75-103)
//   phase 3  closed-loop RK4 / Euler over the horizon                    CK threads, N steps
//            (simulate_rk4_theta_gamma.py:52-68 with the state fed back; velocity
//             transform v_cat = R_theta(theta) R_gamma(-gamma) v when VT == COMPOSE)
//   phase 4  per node: catenary parameter + tension (main_fun.py:418-431, 302-305),
//            augmented-catenary lowest point (main_fun.py:38-111), cost  N*CK items
//   phase 5  J_k = sum_n cost, NaN -> +inf, wave arg-min, outputs        CK threads
//
// Only phase 3 is sequential in n; phases 2 and 4 expose K*N-way parallelism, which is what
// fills the chip at K = 4096 (64 waves' worth of candidates).
#pragma once
#include "device_math.h"

namespace rovmpc {

constexpr int MODEL_BUILTIN = 0;   // saved_models/eq_*.txt rows (complexity 13 / 3), compiled in
constexpr int MODEL_INTERP  = 1;   // any bytecode
constexpr int MODEL_JIT     = 2;   // any bytecode, translated to C++ and compiled with hiprtc at set_model

// Exogenous planes (0..13) the loaded expressions read.  A run-time argument for the interpreter;
// the run-time generated translation unit defines it as a literal, so unused planes vanish there.
#ifndef ROVMPC_JIT_FMAP
#define ROVMPC_JIT_FMAP 0            // feature map of a hiprtc-specialised build (a literal there)
#endif
#ifndef ROVMPC_JIT_N
#define ROVMPC_JIT_N 0               // horizon of a hiprtc-specialised build as a literal (0: run-time)
#endif
#ifndef ROVMPC_JIT_CKC
#define ROVMPC_JIT_CKC 0             // candidates per workgroup of a hiprtc-specialised build as a literal (0: run-time)
#endif
#ifndef ROVMPC_JIT_USED
#define ROVMPC_JIT_USED 0xffffffffu
#endif

// Defined by the run-time generated translation unit (MODEL_JIT only): the two expressions
// over the 18 scaled feature values of one stage.
// e = the model's stage-invariant subexpressions (jit_exo) on the row the stage reads its exogenous slots from: the
// generated code hoists every expensive subtree that no state slot enters (rovmpc.hip::bytecode_to_cxx), so that it is
// evaluated once per row -- end row and midpoint, twice a step -- instead of once per RK4 stage.
#ifndef ROVMPC_JIT_NSUB
#define ROVMPC_JIT_NSUB 0
#endif
// Structure the code generator found in the loaded rows (rovmpc.hip::jit_source; generation-1 map only):
//   ROVMPC_JIT_GI  dgamma/dt reads nothing but x15 / x17 (gamma and its delay slot): the gamma path is candidate-invariant, one
//                  wave integrates it once per workgroup and tabulates what hangs on it, as for the compiled-in rows;
//   ROVMPC_JIT_TS  dtheta/dt reads neither x14 nor x15 (no stage state): the four RK4 slopes of a step are f at the start row,
//                  twice at the midpoint row and at the end row -- no stage loop, and the end row's slope is the next step's
//                  start slope;
//   ROVMPC_JIT_NGSUB  subexpressions of dtheta/dt that read x17 alone (jit_gsub): evaluated on the gamma wave per row.
#ifndef ROVMPC_JIT_GI
#define ROVMPC_JIT_GI 0
#endif
#ifndef ROVMPC_JIT_TS
#define ROVMPC_JIT_TS 0
#endif
#ifndef ROVMPC_JIT_NGSUB
#define ROVMPC_JIT_NGSUB 0
#endif
#ifndef ROVMPC_JIT_PIN_TRIG
#define ROVMPC_JIT_PIN_TRIG 0           // the generated expressions hold a sine: pin its coefficients for the integration loop
#endif
constexpr int JIT_GROW = 16;       // row stride of the gamma table of a ROVMPC_JIT_GI kernel: [0..5] as the compiled-in one, [8 + 3 k + {0, 1, 2}] = g_k at the step's start / midpoint / end row
template <typename T> __device__ void jit_gsub(const T *x, T *g, const Trig<T> &tg);
template <typename T> __device__ void jit_exo(const T *x, T *e, const Trig<T> &tg);
template <typename T> __device__ T jit_f_theta(const T *x, const T *e, const T *g, const Trig<T> &tg);
template <typename T> __device__ T jit_f_gamma(const T *x, const T *e, const Trig<T> &tg);

constexpr int NEXO = 14;           // exogenous feature slots x0..x13 (simply.py:41)
constexpr int NAX  = 8;            // theta axis (x,y) + gamma axis (x,y,z) + unit_rel (x,y,z)

// Scalars of one handle, resident in device memory and read with scalar loads where they are
// used.  Passing them by value as kernel arguments kept ~60 SGPRs live across the whole kernel
// and pushed the sequential phase's loop into SGPR spills (v_writelane/v_readlane per step).

template <typename T> struct RolloutConsts {
    T h, vs_h, inv_h, L, w_per_len, c_lo, c_hi, up, vs, inv_Mm1;       // inv_Mm1 = 1 / (n_shape_pts - 1)
    T w_theta, w_gamma, w_u, w_T, w_taut, rhoL, w_floor, z_floor, theta_ref, gamma_ref;
    T Uref[3];
    T mean[18], inv_scale[18];
};

template <typename T> struct RolloutArgs {
    const T *U;               // [K][N][3]
    const double *state;      // 16 doubles (rovmpc_state)
    const RolloutConsts<T> *k;
    const int32_t *code_th, *code_ga;
    const T *consts;
    const T *Rtab;            // [N][9] (VT_TABLE)
    T *J;                     // [K]
    T *traj_all;              // [K][N+1][2] or null
    double *blk_traj;         // [nblocks][N+1][2]
    int N, K, CK, M, n_th, n_ga, prev_mode, integrator, debug, fmap;
    // sharded step with the library's own collective: GPU-side hand-off of slot row `rank` (null: none).
    // The row may be written once *flag_consumed >= consumed_need; afterwards *flag_rolled = rolled_seq.
    const unsigned long long *flag_consumed;
    unsigned long long *flag_rolled;
    unsigned long long consumed_need, rolled_seq;
    unsigned long long *slot_bad;     // use number of this slot whose row could NOT be written (select then reports NaN)
    unsigned long long handoff_ticks; // give-up time of the hand-off waits, 100 MHz ticks
    unsigned *err;                    // host-mapped error word of the handle (ERR_* bits), read by rovmpc_comm_sync / rovmpc_device_status
    int inject;                       // test hooks (rovmpc_set_option): bit 0 = this launch does not publish its row
    // host-visible completion (rovmpc_mpc_step_sampled): the record is mirrored into pinned, device-mapped host memory
    // (result_host) and the sweeper stores done_seq to *done_flag at system scope once it is out (null: none)
    double *result_host;
    unsigned long long *done_flag;
    unsigned long long done_seq;
    // closed loop: the sweeping workgroup also applies the plant update for the NEXT step (null: no update)
    const double *plant_next;     // 16 doubles, the measured row of step i + 1
    double *plant_state;          // the state the next launch reads
    int plant_feedback;           // 1: keep the model's own (theta, gamma) = first predicted node of this step's winner
    // Closed loop with the state handed over on the GPU (closed_loop_step_kernel; ring == null otherwise).
    // Step g takes P0, P1, V1, A1 from its measured row exo_cur -- known before the loop starts -- and, with feedback and
    // g > 0, (theta0, gamma0, theta_prev, gamma_prev) from ring[g & 3][4], which the sweeper of step g - 1 fills: gamma0 /
    // gamma_prev EARLY (compiled-in model: gamma's path is candidate-invariant, so gamma_1 is known once that sweeper's own
    // gamma chain is through; seq_gamma), theta0 / theta_prev once its arg-min is known (seq_theta).  Everything that does not
    // hang on theta (controls -> LDS, node positions, the gamma table, phase 2b) therefore runs BEFORE the wait.
    const double *exo_cur;
    double *ring;
    unsigned long long *seq_theta, *seq_gamma;
    long long step;               // g
    int from_ring;                // feedback && g > 0: (theta, gamma) come from the ring, else from exo_cur[12..15]
    int wait_theta;               // wait for seq_theta >= g before the theta chain (== from_ring)
    int publish;                  // g + 1 < T: the sweeper publishes for step g + 1
    int NT, nblocks;              // launch geometry (blockDim / gridDim are dependent loads through the implicit arguments)
    int ck_shift;                 // CK == 1 << ck_shift (workgroup sizes are powers of two)
    unsigned used_planes;         // bit s: exogenous plane s is read by the loaded expressions
    unsigned magic_3n;            // floor(2^32 / (3N)) + 1: g / (3N) == umulhi(g, magic) for g < 2^16
    // arg-min epilogue (run by the sweeping workgroup; null result = costs only)
    unsigned long long *granules; // [3][nblocks]: {epoch << 32 | 32 bits} of cost hi, cost lo, winning lane -- the data is the flag
    unsigned epoch;               // launch counter of the handle, never 0: tag of this launch's granules
    int sweeper;                  // workgroup that sweeps: 0 when the grid is one round of workgroups (first dispatched, first
                                  // done, already polling when the stragglers publish), else the last (see argmin_epilogue)
    double *result;               // [5 + 2(N+1)]
    long long *slots;             // [world][R] order-preserving int64 image (sharded step) or null
    long long k_offset;
    int rank, world;
    unsigned long long *stamps;   // diagnostic build only (-DROVMPC_STAMPS): [nblocks][16] 100 MHz ticks
    // Fused sampling (rollout_kernel_sampled, rovmpc_mpc_step_sampled with the compiled-in model): the candidates are DRAWN in
    // phase 0 -- block j of four normals = philox_normal4(seed, step, j), the law of sample_candidates_kernel -- instead of
    // being read: the candidate tensor never exists in HBM, the state arrives as a kernel argument, and the sweeper
    // re-draws the winner's sequence into samp_best (the next step's warm start; the record's u).  Appended at the end
    // of the struct: the other instantiations' argument offsets do not move.
    unsigned long long samp_seed, samp_step;
    double samp_mean[3], samp_std[3];
    const double *samp_warm;      // previous winner's sequence [N][3] (null: no warm start)
    double *samp_best;            // [N][3]: this step's winner's sequence
    double *samp_blk_u;           // [nblocks][3 N]: every workgroup's best candidate's controls, handed over like its trajectory
    double samp_state[ROVMPC_STATE_LEN];
    // Long horizons: the gamma table of the launch, shared between workgroups (null: none).  Workgroup 0 stores its finished
    // table here (write-through) and tags it with the launch epoch; a workgroup that finds the tag when it starts -- the later
    // rounds of a multi-round grid -- loads the table instead of integrating gamma again.  Nobody waits for anybody.
    T *gtab;                      // [8 (N + 1)]
    unsigned long long *gtab_tag;
};

// ---- learned dynamics ---------------------------------------------------------------------

// Bytecode interpreter.  All lanes run the same program (uniform control flow, scalar
// instruction fetch); the operand stack lives in LDS ([depth][lane], conflict-free) with the
// top of stack cached in a register.
template <typename T>
RV_DEV T interp_eval(const int32_t *__restrict__ code, int n, const T *__restrict__ consts,
                     const T *feat, int fstride, T *stack, int sstride) {
    T top = T(0);
    int sp = 0;
    for (int pc = 0; pc < n; ++pc) {
        int ins = code[pc];
        int op = ins & 0xff, arg = ins >> 8;
        switch (op) {
        case ROVMPC_OP_PUSH_C:
            if (sp > 0) stack[(sp - 1) * sstride] = top;
            top = consts[arg]; ++sp; break;
        case ROVMPC_OP_PUSH_F:
            if (sp > 0) stack[(sp - 1) * sstride] = top;
            top = feat[arg * fstride]; ++sp; break;
        case ROVMPC_OP_ADD: top = stack[(sp - 2) * sstride] + top; --sp; break;
        case ROVMPC_OP_SUB: top = stack[(sp - 2) * sstride] - top; --sp; break;
        case ROVMPC_OP_MUL: top = stack[(sp - 2) * sstride] * top; --sp; break;
        case ROVMPC_OP_DIV: top = m_divx(stack[(sp - 2) * sstride], top); --sp; break;
        case ROVMPC_OP_POW: top = m_pow(stack[(sp - 2) * sstride], top); --sp; break;
        case ROVMPC_OP_NEG: top = -top; break;
        case ROVMPC_OP_SIN: top = m_sin(top); break;
        case ROVMPC_OP_COS: top = m_cos(top); break;
        case ROVMPC_OP_TANH: top = m_tanh(top); break;
        case ROVMPC_OP_ABS: top = m_abs(top); break;
        case ROVMPC_OP_SQUARE: top = top * top; break;
        case ROVMPC_OP_EXP: top = m_exp(top); break;
        case ROVMPC_OP_LOG: top = m_log(top); break;
        case ROVMPC_OP_SQRT: top = m_sqrt(top); break;
        case ROVMPC_OP_POWI: {
            int e = arg >= (1 << 23) ? arg - (1 << 24) : arg;     // signed 24-bit
            int ae = e < 0 ? -e : e;
            T b = top, r = T(1);
            while (ae) { if (ae & 1) r *= b; b *= b; ae >>= 1; }
            top = e < 0 ? T(1) / r : r; break;
        }
        case ROVMPC_OP_SAFE_LOG: top = m_log(m_abs(top) + T(1e-5)); break;
        case ROVMPC_OP_SAFE_SQRT: top = m_sqrt(m_abs(top)); break;
        default: top = m_nan<T>(); break;
        }
    }
    return top;
}

// Bits of the handle's error word (host-mapped, written at system scope on the rare failure paths).
constexpr unsigned ERR_WAIT_ROLLED = 1;   // a collective gave up waiting for its rollout's row
constexpr unsigned ERR_CONSUMED    = 2;   // a rollout gave up waiting for the select that frees its slot row
constexpr unsigned ERR_SWEEP       = 4;   // the arg-min sweep gave up waiting for a workgroup's record
RV_DEV void raise_error(unsigned *err, unsigned bit) {
    if (err) __hip_atomic_fetch_or(err, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- arg-min epilogue ----------------------------------------------------------------------
// Agent-scope (sc1, write-through / L1-bypassing) accessors for the bytes one workgroup hands
// to another inside a launch.
RV_DEV void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RV_DEV void st_agent(long long *p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RV_DEV void st_agent(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RV_DEV unsigned long long ld_agent(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The workgroup's best as three tagged granules (epoch << 32 | 32 payload bits): cost high word, cost low word,
// winning lane.  The reader takes a workgroup's record only when all three tags equal this launch's epoch.
// A fourth granule per workgroup says that its best trajectory (write-through stores) has been acknowledged: the cost
// granules go out WITHOUT waiting for that drain, so the sweep sees a workgroup's cost ~0.7 us earlier, and only the
// winner's trajectory flag is waited for (it is almost always up by then).
constexpr int GRAN = 4;
RV_DEV void publish_traj_ready(unsigned long long *granules, int nblocks, unsigned epoch) {
    st_agent(granules + 3 * (size_t)nblocks + blockIdx.x, (unsigned long long)epoch << 32);
}
RV_DEV void publish_best(unsigned long long *granules, int nblocks, unsigned epoch, double cost, unsigned lane) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(cost), tag = (unsigned long long)epoch << 32;
    st_agent(granules + blockIdx.x, tag | (bits >> 32));
    st_agent(granules + nblocks + blockIdx.x, tag | (bits & 0xffffffffULL));
    st_agent(granules + 2 * (size_t)nblocks + blockIdx.x, tag | lane);
}
RV_DEV double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RV_DEV float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RV_DEV void st_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RV_DEV long long ld_agent(const long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One workgroup: lexicographic (cost, index) minimum over the per-block bests, then the
// result record [J*, k*, u(3), (theta,gamma)_0..N].  If slots != null also writes the
// order-preserving int64 image of the record into slots[rank][*] and INT64_MAX elsewhere
// (input of the single all-reduce(min) of the candidate-sharded step).  `scratch` = 16
// doubles of LDS.
template <typename T, bool LEAN = false, bool SAMPLE = false>
RV_DEV void argmin_epilogue(const RolloutArgs<T> &a, const unsigned long long *granules, const double *blk_traj, const T *U,
                            double *result, double *scratch) {
    const int nblocks = a.nblocks, N = a.N, CK = a.CK, NT = a.NT, rank = a.rank, world = a.world;
    long long *slots = a.slots;
    const unsigned epoch = a.epoch;
    double *sJ = scratch;                                    // [8]
    long long *sK = reinterpret_cast<long long *>(scratch + 8);   // [8]
    const int tid = threadIdx.x, nw = (NT + 63) >> 6;
    double Jd = __builtin_inf();
    long long kk = 0x7fffffffffffffffLL;
    auto take = [&](double oJ, long long ok) {
        const bool better = (oJ < Jd) | ((oJ == Jd) & (ok < kk));
        Jd = better ? oJ : Jd; kk = better ? ok : kk;
    };
    // Sweep (cdna_hip_programming.md G16, form R2: the data is the flag): every workgroup publishes its best as
    // three 8-byte granules tagged with this launch's epoch, after draining the write-through stores of its
    // trajectory.  This workgroup (number 0 when the whole grid is resident at once; otherwise the LAST one of the
    // grid, dispatched last, so that it never sits on a CU slot while earlier rounds of workgroups still queue for
    // one -- measured: workgroup 0 as the sweeper cost a whole round at C3's two-round launch) re-reads, with agent-scope loads, the granules it has not yet seen complete, until none
    // is pending -- no ticket counter, no fence, and the other workgroups leave as soon as they have published.
    {
        int j = 0;
        // 60 s of the 100 MHz clock; in the closed loop with GPU-side hand-off the hand-off timeout (both grids in flight
        // have to drain)
        const unsigned long long give_up = wall_clock64() + (a.ring ? a.handoff_ticks : 6000000000ULL);
        for (unsigned it = 1;; ++it) {
            bool pending = false;
            for (;;) {
                const int b = tid + j * NT;
                if (b >= nblocks) break;
                const unsigned long long g0 = ld_agent(granules + b), g1 = ld_agent(granules + nblocks + b),
                                         g2 = ld_agent(granules + 2 * (size_t)nblocks + b);
                if ((unsigned)(g0 >> 32) != epoch || (unsigned)(g1 >> 32) != epoch || (unsigned)(g2 >> 32) != epoch) { pending = true; break; }
                const double cost = __longlong_as_double((long long)((g0 << 32) | (g1 & 0xffffffffULL)));
                take(cost, (long long)b * CK + (long long)(g2 & 0xffffffffULL));
                ++j;
            }
            if (!__syncthreads_or(pending)) break;
            if ((it & 1023u) == 0 && __syncthreads_or(wall_clock64() > give_up)) {   // some workgroup never published: a NaN cost says so
                Jd = __builtin_nan(""); kk = 0;
                if (tid == 0) raise_error(a.err, ERR_SWEEP);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    for (int off = 32; off > 0; off >>= 1) take(__shfl_down(Jd, off, 64), __shfl_down(kk, off, 64));
    if ((tid & 63) == 0) { sJ[tid >> 6] = Jd; sK[tid >> 6] = kk; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < nw; ++w)
            if (sJ[w] < Jd || (sJ[w] == Jd && sK[w] < kk)) { Jd = sJ[w]; kk = sK[w]; }
        sK[0] = kk;
        sJ[0] = Jd;
    }
    __syncthreads();
    const long long kbest = sK[0];
    const double Jbest = sJ[0];
    const int R = 5 + 2 * (N + 1);
    const double *bt = blk_traj + (size_t)(kbest / CK) * (N + 1) * 2;
    if (Jbest == Jbest) {
        // the winner's trajectory must be out (its flag granule; the cost granules did not wait for it)
        if (tid == 0) {
            const unsigned long long *f = granules + 3 * (size_t)nblocks + (size_t)(kbest / CK);
            const unsigned long long give_up = wall_clock64() + (a.ring ? a.handoff_ticks : 6000000000ULL);
            while ((unsigned)(ld_agent(f) >> 32) != epoch) {
                if (wall_clock64() > give_up) { raise_error(a.err, ERR_SWEEP); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    // (sharded closed loop: the next step's theta is the GLOBAL winner's -- select_kernel hands it over behind the all-reduce)
    if (!LEAN && a.ring && a.publish && !slots && tid < 64) {
        // closed loop, GPU-side hand-off: the next step's (theta0, gamma0, theta_prev, gamma_prev) = nodes 1 and 0 of the
        // winner, before anything else -- the record below is off the loop's critical path
        if (a.plant_feedback && tid < 4) st_agent(&a.ring[(int)((a.step + 1) & 3) * 4 + tid], ld_agent(&bt[tid < 2 ? 2 + tid : tid - 2]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) st_agent(a.seq_theta, (unsigned long long)(a.step + 1));
    }
    if (SAMPLE) {
        // the winner's control sequence (its workgroup handed it over with the trajectory): the record's u and the next
        // step's warm start
        const double *bu = a.samp_blk_u + (size_t)(kbest / CK) * 3 * N;
        for (int j = tid; j < 3 * N; j += NT) a.samp_best[j] = ld_agent(&bu[j]);
        __syncthreads();
    }
    bool row_free = true;
    if (!LEAN && a.flag_consumed) {
        // the slot buffer is reused every few steps: its previous contents must have been read by that step's select.
        // If that never happens (a failed collective) the row is NOT rewritten: the error word and the slot's
        // bad-use mark make this step's global record a NaN and rovmpc_comm_sync an error.
        int *s_ok = reinterpret_cast<int *>(scratch + 15);     // sK[7]: read by thread 0 only, before the last barrier
        if (tid == 0) {
            const unsigned long long give_up = wall_clock64() + a.handoff_ticks;
            bool ok = true;
            while (ld_agent(a.flag_consumed) < a.consumed_need) {
                if (wall_clock64() > give_up) { ok = false; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok) { raise_error(a.err, ERR_CONSUMED); st_agent(a.slot_bad, a.rolled_seq); }
            *s_ok = ok ? 1 : 0;
        }
        __syncthreads();
        row_free = *s_ok != 0;
    }
    for (int i = tid; i < R; i += NT) {
        double v;
        if (i == 0) v = row_free ? Jbest : __builtin_nan("");
        else if (i == 1) v = (double)(kbest + a.k_offset);
        else if (i < 5) v = SAMPLE ? a.samp_best[i - 2] : (double)U[(size_t)kbest * N * 3 + (i - 2)];
        else v = ld_agent(&bt[i - 5]);
        result[i] = v;
        if (!LEAN && a.result_host) a.result_host[i] = v;
        if (!LEAN && slots && row_free) st_agent(&slots[(size_t)rank * R + i], ordered_key(v));
    }
    if (!LEAN && slots && row_free) {
        for (int i = tid; i < world * R; i += NT)
            if (i / R != rank) st_agent(&slots[i], 0x7fffffffffffffffLL);
    }
    if (!LEAN && a.done_flag) {
        // the record went to host memory: drain every wave's stores, then release the sequence number the host spins on
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (!LEAN && a.flag_rolled && !(a.inject & 1)) {
        // publish: the row went out write-through at agent scope; once every wave's stores are acknowledged the
        // sequence number follows (the collective stream's wait kernel polls it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) st_agent(a.flag_rolled, a.rolled_seq);
    }
    // Closed loop on one GPU: the plant update of the next step (plant_update_kernel's rule) rides on this
    // workgroup -- every other workgroup has finished, nobody reads the state any more.
    if (!LEAN && a.plant_next && !(a.ring && slots) && tid < 16) {
        const double *plant_next = a.plant_next;
        double *plant_state = a.plant_state;
        if (!a.plant_feedback) {
            plant_state[tid] = plant_next[tid];
        } else if (tid < 12) {
            plant_state[tid] = plant_next[tid];
        } else if (tid == 12) {
            // (theta, gamma) of this step = node 0 of the winner's trajectory, of the next = node 1
            const double th = a.ring ? ld_agent(&bt[0]) : plant_state[12], ga = a.ring ? ld_agent(&bt[1]) : plant_state[13];
            plant_state[14] = th; plant_state[15] = ga;
            plant_state[12] = ld_agent(&bt[2]); plant_state[13] = ld_agent(&bt[3]);
        }
    }
}

// ---- the kernel ---------------------------------------------------------------------------

// A theta slot somebody waits for holds this marker (a NaN payload no computation produces) until the integrating wave
// stores the node: relaxed 32/64-bit LDS accesses, single-copy atomic.
RV_DEV void theta_slot_mark(double *p) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), 0x7ff85ea71e5007e7ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RV_DEV void theta_slot_mark(float *p) { __hip_atomic_store(reinterpret_cast<unsigned int *>(p), 0x7fc5ea71u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RV_DEV bool theta_slot_marked(double *p) { return __hip_atomic_load(reinterpret_cast<unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0x7ff85ea71e5007e7ull; }
RV_DEV bool theta_slot_marked(float *p) { return __hip_atomic_load(reinterpret_cast<unsigned int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0x7fc5ea71u; }

// In-kernel phase stamps exist only in the diagnostic library (make diag, -DROVMPC_STAMPS); the
// product library contains none of this code.
#ifdef ROVMPC_STAMPS
#define RV_STAMP(i) do { if (threadIdx.x == 0 && a.stamps) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)
#define RV_STAMP_W(i) do { if ((threadIdx.x & 63) == 0 && a.stamps) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)   // lane 0 of the calling wave
// slot i <- SIMD of every wave of the workgroup: nibble w = 8 | SIMD_ID of wave w (HW_ID bits 5:4)
#define RV_STAMP_SIMD(i) do { if ((threadIdx.x & 63) == 0 && a.stamps) atomicOr(&a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)], \
    (unsigned long long)(8u | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3u)) << (4 * (threadIdx.x >> 6))); } while (0)
// slot i <- HW_ID (hwreg 4: wave, simd, pipe, cu, sh, se) | XCC_ID (hwreg 20) << 32: which CU ran the workgroup
#define RV_STAMP_HW(i) do { if (threadIdx.x == 0 && a.stamps) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = \
    (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } while (0)
#else
#define RV_STAMP(i) do { } while (0)
#define RV_STAMP_HW(i) do { } while (0)
#define RV_STAMP_SIMD(i) do { } while (0)
#define RV_STAMP_W(i) do { } while (0)
#endif

// LDS plane addressing: plane p, node n (0..N), lane c (0..CK-1); c fastest => conflict-free.
#define RV_PL(base, p, n, c) (base)[((p) * (N + 1) + (n)) * CK + (c)]
// exogenous feature plane s (0..13): a hiprtc-specialised kernel keeps only the planes its expressions read, packed
// (xpl(s) = number of used planes below s, a literal after unrolling); the other variants keep plane s at s
#define RV_PX(s, n, c) RV_PL(sX, xpl(s), n, c)

// Node planes kept in LDS.  The compiled-in model reads only x3 (and x14..x17, which are
// state), so its workgroups keep P (3), theta/gamma (2) and either the x3 plane or the five
// rotation-axis components; the interpreter keeps the whole scaled exogenous row.
__host__ __device__ inline int popcount14(unsigned used) {
    int n = 0;
    for (int s = 0; s < NEXO; ++s) n += (used >> s) & 1u;
    return n;
}
// Planes a hiprtc-specialised kernel keeps in LDS: the used ones, minus -- under VT_COMPOSE -- those that hang on the
// velocity (planes 3..8 and 13 of generations 1/2, every plane of generation 3): only the integrating lane reads them,
// from registers (integrate_jit / integrate_dd_jit).
__host__ __device__ inline unsigned jit_lds_planes(unsigned used, int vt, int fmap) {
    if (vt != ROVMPC_VT_COMPOSE) return used;
    return fmap == ROVMPC_FEATURES_GEN3 ? 0u : used & ~0x21f8u;
}
__host__ __device__ inline int rollout_nx(int model, int vt, unsigned used = 0xffffffffu) {
    if (model == MODEL_BUILTIN) return vt == ROVMPC_VT_COMPOSE ? 0 : 1;
    return model == MODEL_JIT ? (popcount14(used) > 0 ? popcount14(used) : 1) : NEXO;
}
__host__ __device__ inline int rollout_na(int model, int vt) {
    return vt == ROVMPC_VT_COMPOSE ? NAX : 5;   // axes are reused by phase 4b; compose: + 3 (unit_rel, or w of the compiled-in path)
}
constexpr int HDR = 48;            // header: flags (8 slots) + mean[18] + inv_scale[18] (+ pad)

template <typename T> __host__ __device__ inline size_t rollout_lds_elems(int N, int CK, int model, int vt, unsigned used = 0xffffffffu, int jit_gi = 0) {
    size_t planes = 3 /*P*/ + 2 /*theta,gamma*/ + rollout_nx(model, vt, used) + rollout_na(model, vt);
    size_t e = HDR;
    e += planes * (size_t)(N + 1) * CK;              // node planes
    e += (size_t)CK * ((3 * N) | 1);                 // U chunk, odd row stride (bank spread)
    e += (size_t)CK * N;                             // node costs
    e += (size_t)2 * CK * N;                         // warm start of the second catenary solve of a node (u, exp u)
    if (model == MODEL_INTERP) e += (size_t)(18 + ROVMPC_MAX_STACK) * CK;   // features + stack
    if (model == MODEL_BUILTIN) e += (size_t)8 * (N + 1);                    // candidate-invariant gamma table
    if (model == MODEL_JIT && jit_gi) e += (size_t)JIT_GROW * (N + 1);       // the same of a loaded model with a candidate-invariant gamma path
    return e;
}

// LEAN: the plain single-problem step (rovmpc_step_device and the host-pointer entry points): one problem, no slot image,
// no hand-off flags, no host mirror, no plant update -- those branches are compiled out (measured on one box: the full
// kernel is 0.3 us slower per C2 step than the round-1 kernel, which had none of them).
// LONGH: the long-horizon instance (3 N + 2 > 64, compiled-in model, plain launches): shared gamma sines, the gamma table
// shared between workgroups, geometry waves chasing the theta wave.  A separate instantiation, so that none of that code sits
// in the instruction stream of the C2-sized kernels (their layout is touchy: +0.15 us at C2 with the code merely present).
template <typename T, int MODEL, int VT, bool HANDOFF = false, bool LEAN = false, bool SAMPLE = false, bool LONGH = false, int CKC = 0, int NC = 0>
RV_DEV void rollout_body(const RolloutArgs<T> &a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    // CKC: the candidates per workgroup as a literal (16: the single-problem step) -- row strides become immediates of the
    // LDS instructions, one address register per plane is bumped once per trip of the theta chain's loop
    // NC: the horizon as a literal too -- then every plane of the LDS layout sits at an immediate offset of one address
    const int N = NC ? NC : a.N, CK = CKC ? CKC : a.CK, K = a.K;
    const int cks = CKC == 16 ? 4 : a.ck_shift, ckm = CK - 1;        // i / CK == i >> cks, i % CK == i & ckm
    const unsigned used = MODEL == MODEL_JIT ? (unsigned)ROVMPC_JIT_USED : a.used_planes;
    const int fmap = MODEL == MODEL_JIT ? (int)ROVMPC_JIT_FMAP : a.fmap;
    auto uses = [&](int plane) { return (used >> plane) & 1u; };
    // The handle's scalars live in device memory that nothing writes while a kernel runs.  fp32: read through the CONSTANT
    // address space they become scalar loads (s_load, SGPR results, placed where they are used) instead of the vector loads
    // the compiler must issue for memory it cannot prove unclobbered across the kernel's LDS and global stores (C3:
    // 52.5 -> 51.8 us).  fp64: the same change costs more in spilled SGPRs than the loads cost (42 -> 86 spill slots,
    // B = 64: 371 -> 380 us, measured on one box), so the fp64 kernels keep the plain pointer.
    typedef const RolloutConsts<T> __attribute__((address_space(4))) *ConstsPtrK;
    typedef const RolloutConsts<T> *ConstsPtrG;
    typedef typename RvCond<sizeof(T) == 4, ConstsPtrK, ConstsPtrG>::type ConstsPtr;
    const auto &kk = *(ConstsPtr)a.k;
    const int tid = threadIdx.x, NT = a.NT;
    const int k0 = blockIdx.x * CK;
    // Batched launch (rovmpc_step_batch_device): blockIdx.y = problem.  Every per-problem array is the single-problem
    // array repeated B times; a problem's workgroups, granules, sweeper and record never touch another problem's.
    const int prob = LEAN ? 0 : (int)blockIdx.y;
    const T *Ub = a.U + (size_t)prob * K * N * 3;
    const int nvalid = min(CK, K - k0);
    const unsigned used_lds = MODEL == MODEL_JIT ? jit_lds_planes(used, VT, fmap) : used;
    const int NX = rollout_nx(MODEL, VT, used_lds);
    auto xpl = [&](int s) { return MODEL == MODEL_JIT ? popcount14(used_lds & ((1u << s) - 1u)) : s; };
    constexpr int NA = VT == ROVMPC_VT_COMPOSE ? NAX : 5;

    // carve LDS
    int *s_best_c = reinterpret_cast<int *>(smem);   // header
    T *sMean = smem + 8, *sInv = smem + 26;          // scaler constants (lane-uniform reads)
    const int US = (3 * N) | 1;                      // padded U row stride
    T *sP = smem + HDR;                              // 3 planes
    T *sY = sP + 3 * (N + 1) * CK;                   // theta, gamma planes
    T *sX = sY + 2 * (N + 1) * CK;                   // NX planes (scaled features)
    T *sA = sX + NX * (N + 1) * CK;                  // NA planes (rotation axes [, unit_rel])
    T *sU = sA + NA * (N + 1) * CK;                  // [c][n][3]
    T *sC = sU + CK * US;                            // [n][c] node costs
    T *sW = sC + CK * N;                             // [2][n][c]: root of phase 4a's solve and its exp (warm start of 4b's)
    T *sF = sW + 2 * CK * N;                         // interpreter: 18 feature rows + stack
    T *sG = sW + 2 * CK * N;                             // compiled-in model: gamma table [N + 1][8]
    int *s_prog = s_best_c + 2;                      // [1]: theta steps finished (early phase-4b batch)

    RV_STAMP(0);
    // state: declared here (the gamma lambdas capture it), loaded after the controls' loads are in flight
    T P0x, P0y, P0z, V0x, V0y, V0z, A0x, A0y, A0z, th0, ga0, thm0, gam0;
    // (closed loop with GPU-side hand-off: the measured row of this step stands in for the state; theta / gamma arrive
    // later, in the waves that need them -- ring_wait / ring_get)
    const double *sdg = SAMPLE ? a.samp_state : (HANDOFF ? a.exo_cur : a.state + (size_t)prob * ROVMPC_STATE_LEN);
    auto sd_at = [&](int i) -> double { return sdg[i]; };
    // Whole-wave call: lane 0 polls the sequence word until the record of this step is out (bounded: on giving up it raises
    // the error word and the wave goes on with whatever the ring holds).
    auto ring_wait = [&](const unsigned long long *seq) {
        if ((tid & 63) == 0) {
            const unsigned long long give_up = wall_clock64() + a.handoff_ticks;
            while (ld_agent(seq) < (unsigned long long)a.step) {
                if (wall_clock64() > give_up) { raise_error(a.err, ERR_SWEEP); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    };
    auto ring_get = [&](int idx) -> T {                       // wave-uniform value
        const double v = ld_agent(&a.ring[(a.step & 3) * 4 + idx]);
        const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(__double_as_longlong(v) & 0xffffffffLL));
        const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)__double_as_longlong(v) >> 32));
        return (T)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };

    // Compiled-in model: dgamma/dt = x15 - x17 reads gamma and its delay slot only -- no control, no
    // theta -- so the gamma path is the SAME for every candidate and needs nothing but the state.  One
    // wave (the "gamma wave") integrates all N steps (gamma_chain) and fills the per-step table the
    // theta chain and the velocity transform read (gamma_sines) while the other waves are in phase 2a:
    //   row n: [0] sin gamma_n, [1] cos gamma_n, [2] G_n, [3] sin(x17 at t_n+1), [4] sin(x17 at the
    //   midpoint), [5] gamma_{n+1};  row N: [0] sin(x17 at t_0).
    // G_n is the candidate-invariant part of the RK4 sum of dtheta/dt (see theta_path).
    const int nintB = ((CK + 15) / 16) * 64;                 // theta waves of the compiled-in model
    const bool wideB = MODEL == MODEL_BUILTIN && NT >= nintB + 64;
    // (a loaded model whose gamma path is candidate-invariant -- ROVMPC_JIT_GI -- gets a gamma wave as well: the wave behind its
    // single integrating wave)
    constexpr bool JGI = MODEL == MODEL_JIT && ROVMPC_JIT_GI != 0;
    constexpr int GROW = JGI ? JIT_GROW : 8;                 // row stride of the gamma table
    const int nintG = JGI ? 64 : nintB;
    const bool wideG = (MODEL == MODEL_BUILTIN || JGI) && NT >= nintG + 64;
    const bool gwave = (MODEL == MODEL_BUILTIN || JGI) && (wideG ? (tid >= nintG && tid < nintG + 64) : tid < 64);
    auto gamma_chain = [&]() {
        const int lane = tid & 63;
        const Trig<T> trigj(false);               // (loaded models: the context their generated expressions take)
        const int nsteps = (a.debug & 1) ? 0 : N;
        const T m15 = sMean[15], i15 = sInv[15], m17 = sMean[17], i17 = sInv[17];
        const T hstep = kk.h, hh = T(0.5) * hstep, h6 = hstep / T(6);
        // ~20 dependent operations per step in the reference's own order (the recurrence amplifies rounding,
        // so the affine closed form of the step -- two FMAs -- does not hold 1e-9 beyond ~30 steps: measured),
        // carried by every lane, stored by lane 0.  One loop per mode: no selects on the chain.
        T ga = ga0, gam = gam0;
        // dgamma/dt on (scaled x15, scaled x17): the compiled-in row, or the loaded one (which reads nothing else)
        auto fgam = [&](T s15, T p17) -> T {
            if (JGI) {
                T x[18];
#pragma unroll
                for (int sl = 0; sl < 18; ++sl) x[sl] = T(0);
                x[15] = s15; x[17] = p17;
                T e[ROVMPC_JIT_NSUB > 0 ? ROVMPC_JIT_NSUB : 1];
                jit_exo<T>(x, e, trigj);                       // (subexpressions of constants only; the others are dead here)
                return jit_f_gamma<T>(x, e, trigj);
            }
            return s15 - p17;
        };
        // Horizons of up to 64 steps can keep the path in registers -- lane n holds gamma_{n+1} -- and store it once behind
        // the loop: a masked store and its two jumps in every step of the chain cost more than three selects; with two steps
        // per trip, the second on the first one's names, there is no copy of gamma_n on the chain either (C2: -0.4 us).  As
        // a literal of the loop only: the plain short-horizon instances always (N <= 20 there), every other one through a
        // second copy of the loop (C3, literal N = 50: -0.9 us).
        T kept = T(0);
        auto run = [&](auto HOLD, auto EULER, auto KEEP) {
            auto one = [&](int n, const T gam, const T ga) -> T {
                if (!JGI) {
                    // The compiled-in row dgamma/dt = s15 - p17 with every fused multiply-add spelt out (the recurrence amplifies
                    // rounding: left to the compiler's contraction, two instances of this loop need not round alike)
                    const T s17a = (gam - m17) * i17, d17 = ga - m17, s17b = d17 * i17;   // np.roll delay slot, simply.py:35-38
                    const T p17m = HOLD.value ? s17a : m_fma(i17, d17, s17a) * T(0.5);
                    const T p17e = HOLD.value ? s17a : s17b;
                    const T k1g = m_fma(i15, ga - m15, -s17a);
                    T gan;
                    if (EULER.value) {
                        gan = m_fma(k1g, hstep, ga);                                // main_fun.py:762
                    } else {
                        const T k2g = m_fma(i15, m_fma(hh, k1g, ga) - m15, -p17m);
                        const T k3g = m_fma(i15, m_fma(hh, k2g, ga) - m15, -p17m);
                        const T k4g = m_fma(i15, m_fma(hstep, k3g, ga) - m15, -p17e);
                        gan = m_fma(h6, m_fma(T(2), k3g, m_fma(T(2), k2g, k1g)) + k4g, ga);   // :66
                    }
                    if (KEEP.value) { if (lane == n) kept = gan; }
                    else if (lane == 0) sG[GROW * n + 5] = gan;
                    return gan;
                }
                const T s17a = (gam - m17) * i17, s17b = (ga - m17) * i17;      // np.roll delay slot, simply.py:35-38
                const T p17m = HOLD.value ? s17a : (s17a + s17b) / T(2);
                const T p17e = HOLD.value ? s17a : s17b;
                const T k1g = fgam((ga - m15) * i15, s17a);
                T gan;
                if (EULER.value) {
                    gan = ga + k1g * hstep;                                     // main_fun.py:762
                } else {
                    const T k2g = fgam(((ga + hh * k1g) - m15) * i15, p17m);
                    const T k3g = fgam(((ga + hh * k2g) - m15) * i15, p17m);
                    const T k4g = fgam(((ga + hstep * k3g) - m15) * i15, p17e);
                    gan = ga + h6 * (k1g + T(2) * k2g + T(2) * k3g + k4g);      // :66
                    if (JGI && !ROVMPC_JIT_TS && lane == 0) {      // gamma's stage states, for a dtheta/dt that reads them (slots 2..4 of the row)
                        sG[GROW * n + 2] = ga + hh * k1g; sG[GROW * n + 3] = ga + hh * k2g; sG[GROW * n + 4] = ga + hstep * k3g;
                    }
                }
                if (KEEP.value) { if (lane == n) kept = gan; }
                else if (lane == 0) sG[GROW * n + 5] = gan;
                return gan;
            };
            int n = 0;
            if (KEEP.value) {
                for (; n + 1 < nsteps; n += 2) {
                    const T g1 = one(n, gam, ga);
                    const T g2 = one(n + 1, ga, g1);
                    gam = g1; ga = g2;
                }
            }
            for (; n < nsteps; ++n) { const T g1 = one(n, gam, ga); gam = ga; ga = g1; }
            if (KEEP.value && lane < nsteps) sG[GROW * lane + 5] = kept;
        };
        auto run_mode = [&](auto KEEP) {
            const bool hold = a.prev_mode == ROVMPC_PREV_HOLD;
            if (a.integrator == ROVMPC_EULER) run(BoolC<false>{}, BoolC<true>{}, KEEP);
            else if (hold) run(BoolC<true>{}, BoolC<false>{}, KEEP);
            else run(BoolC<false>{}, BoolC<false>{}, KEEP);
        };
        constexpr bool CAN_KEEP = MODEL == MODEL_BUILTIN || (JGI && ROVMPC_JIT_TS != 0);   // (loaded rows whose dtheta/dt reads gamma's stage states store those per step anyway: measured, no gain)
        if (CAN_KEEP && MODEL == MODEL_BUILTIN && !HANDOFF && !SAMPLE && (!LONGH || (NC > 0 && NC <= 64))) run_mode(BoolC<true>{});
        else if (CAN_KEEP && N <= 64) run_mode(BoolC<true>{});
        else run_mode(BoolC<false>{});
    };
    // G_n = sinA + 4 sinM + sinE (RK4; sinA for Euler), sinA_n = sin(x17 at t_n) = row n-1's [3]
    auto gamma_G = [&](int first, int stride) {
        const int nsteps = (a.debug & 1) ? 0 : N;
        const bool hold = a.prev_mode == ROVMPC_PREV_HOLD, euler = a.integrator == ROVMPC_EULER;
        for (int n = first; n < nsteps; n += stride) {
            const T sinA = n == 0 ? sG[8 * N] : sG[8 * (n - 1) + 3];
            const T sinM = hold ? sinA : sG[8 * n + 4], sinE = hold ? sinA : sG[8 * n + 3];
            sG[8 * n + 2] = euler ? sinA : (sinA + sinE) + T(4) * sinM;
        }
    };
    // (first, stride): the items this thread evaluates -- the gamma wave's lanes alone, or, for long horizons, every thread of
    // the workgroup once the chain is through (share_sines below); with_G: the G_n pass follows at once (it reads every sine:
    // shared sines leave it to phase 2b, behind the barrier)
    auto gamma_sines = [&](int first, int stride, bool with_G) {
        const int nsteps = (a.debug & 1) ? 0 : N;
        const Trig<T> trig(LONGH);       // (short horizons: one evaluation per lane -- pinning the constants costs more moves than it saves)
        const T m17 = sMean[17], i17 = sInv[17];
        // item 3n + r -- r = 0 sincos(gamma_n), 1 sin(x17 at t_n+1), 2 sin(x17 at the midpoint); the last
        // item is sin(x17 at t_0).  One wave's DS operations complete in order, so the reads see the
        // chain's stores.
        // (and one more: sincos(gamma_N), which only the geometry of the last node reads -- row N, slots 2 / 3)
        // (no branch in the item: clamped reads and selects; a sine item's unused cosine goes to a free slot of its row, [6] / [7])
        for (int i = first; i <= 3 * nsteps + (nsteps > 0 ? 1 : 0); i += stride) {
            const bool last = i == 3 * nsteps + 1, t0 = i == 3 * nsteps;
            const int n = last ? nsteps : (t0 ? 0 : i / 3), r = last ? 0 : (t0 ? 3 : i - 3 * n);
            const T r1 = sG[8 * max(n - 1, 0) + 5], r2 = sG[8 * max(n - 2, 0) + 5];
            const T g_n = n == 0 ? ga0 : r1;
            const T g_m = n == 0 ? gam0 : (n == 1 ? ga0 : r2);
            const T s17a = (g_m - m17) * i17, s17b = (g_n - m17) * i17;
            T sv, cv;
            trig.sincos(r == 0 ? g_n : (r == 1 ? s17b : (r == 2 ? (s17a + s17b) / T(2) : s17a)), &sv, &cv);
            // sine: row n [0] (sincos gamma_n), [3] / [4] (x17 at the step's end / midpoint), row N [0] (x17 at t_0), row N [2] (gamma_N)
            const int at_s = last ? 8 * n + 2 : (t0 ? 8 * N : 8 * n + (r == 0 ? 0 : 2 + r));
            const int at_c = last ? 8 * n + 3 : (t0 ? 8 * N + 6 : 8 * n + (r == 0 ? 1 : 5 + r));
            sG[at_s] = sv; sG[at_c] = cv;
        }
        if (with_G) gamma_G(first, stride);
    };
    // Loaded model with a candidate-invariant gamma path (ROVMPC_JIT_GI): what the integrating wave and the geometry read of it.
    // Row n of the table: [0], [1] sincos(gamma_n) (n = 0..N); [5] gamma_{n+1} (the chain); [8 + 3 k + w] the x17-only
    // subexpression k of dtheta/dt (jit_gsub) on the delay slot at the start (w = 0), midpoint (1) and end (2) row of step n.
    auto gamma_table_jit = [&]() {
        const int lane = tid & 63;
        const Trig<T> trigj(false);
        const int nsteps = (a.debug & 1) ? 0 : N;
        const bool hold = a.prev_mode == ROVMPC_PREV_HOLD;
        const T m17 = sMean[17], i17 = sInv[17];
        constexpr int NG = ROVMPC_JIT_NGSUB;
        const int n_sub = NG > 0 ? 3 * nsteps : 0;
        for (int i = lane; i < n_sub + (nsteps > 0 ? N + 1 : 0); i += 64) {
            if (i < n_sub) {
                const int n = i / 3, w = i - 3 * n;
                const T g_n = n == 0 ? ga0 : sG[GROW * (n - 1) + 5];
                const T g_m = n == 0 ? gam0 : (n == 1 ? ga0 : sG[GROW * (n - 2) + 5]);
                const T s17a = (g_m - m17) * i17, s17b = (g_n - m17) * i17;
                T x[18];
#pragma unroll
                for (int sl = 0; sl < 18; ++sl) x[sl] = T(0);
                x[17] = (hold || w == 0) ? s17a : (w == 1 ? (s17a + s17b) / T(2) : s17b);
                T g[NG > 0 ? NG : 1];
                jit_gsub<T>(x, g, trigj);
#pragma unroll
                for (int k = 0; k < NG; ++k) sG[GROW * n + 8 + 3 * k + w] = g[k];
            } else {
                const int n = i - n_sub;
                T sv, cv;
                m_sincos(n == 0 ? ga0 : sG[GROW * (n - 1) + 5], &sv, &cv);
                sG[GROW * n] = sv; sG[GROW * n + 1] = cv;
            }
        }
    };
    // sines shared by the whole workgroup: when they are more than one pass of the gamma wave (3 N + 2 > 64) and phase 2 is
    // bounded by that wave (C3, N = 50: chain 3.2 us + three passes 1.8 us against 1.3 us of phase 2a) the other waves, idle
    // at the barrier, take their share as soon as the chain is through (LDS flag): one pass instead of three
    const bool share_sines = LONGH && MODEL == MODEL_BUILTIN && wideB && 3 * N + 2 > 64;
    int *s_chain_done = s_best_c + 6;

    // ---- phase 0: candidate controls -> LDS, coalesced ------------------------------------
    // (compiled-in model, wide workgroup: the gamma wave has nothing to fetch)
    const int ltid = wideG ? (tid < nintG ? tid : tid - 64) : tid, LNT = wideG ? NT - 64 : NT;
    if (!(wideG && gwave)) {
        const T *src = Ub + (size_t)k0 * N * 3;
        const int tot = nvalid * N * 3;
        constexpr int VW = 16 / sizeof(T);           // elements per 16-byte lane load
        const bool vec = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (tot % VW == 0);
        if (SAMPLE) {
            // draw the chunk: element e = (k N + n) 3 + c of the candidate tensor is mean[c] + std[c] z_e; a lane takes whole
            // Philox blocks (four consecutive elements) and keeps those that fall into this workgroup's chunk
            const long long e_lo = (long long)k0 * N * 3, e_hi = e_lo + tot;
            const long long jb_lo = e_lo >> 2, jb_hi = (e_hi + 3) >> 2;
            for (long long jb = jb_lo + ltid; jb < jb_hi; jb += LNT) {
                double z[4];
                philox_normal4(a.samp_seed, a.samp_step, jb, z);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long e = 4 * jb + q;
                    if (e < e_lo || e >= e_hi) continue;
                    const int g = (int)(e - e_lo), c = (int)__umulhi((unsigned)g, a.magic_3n), j = g - c * (3 * N), cc = j % 3;
                    T v = (T)::fma(a.samp_std[cc], z[q], a.samp_mean[cc]);
                    if (a.samp_warm && e < 3 * N) {          // candidate 0: the previous winner shifted by one step
                        const int n = j / 3, sn = n + 1 < N ? n + 1 : N - 1;
                        v = (T)a.samp_warm[3 * sn + cc];
                    }
                    sU[c * US + j] = v;
                }
            }
            for (int i = tot + ltid; i < CK * N * 3; i += LNT) {
                const int c = (int)__umulhi((unsigned)i, a.magic_3n), j = i - c * (3 * N);
                sU[c * US + j] = T(0);
            }
        } else if (vec) {
            typedef T vecT __attribute__((ext_vector_type(VW)));
            const vecT *src4 = reinterpret_cast<const vecT *>(src);
            for (int i = ltid; i < tot / VW; i += LNT) {
                const vecT v = src4[i];
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    const int g = i * VW + e, c = (int)__umulhi((unsigned)g, a.magic_3n), j = g - c * (3 * N);
                    sU[c * US + j] = v[e];
                }
            }
            for (int i = tot + ltid; i < CK * N * 3; i += LNT) {
                const int c = (int)__umulhi((unsigned)i, a.magic_3n), j = i - c * (3 * N);
                sU[c * US + j] = T(0);
            }
        } else {
            for (int i = ltid; i < CK * N * 3; i += LNT) {
                const int c = (int)__umulhi((unsigned)i, a.magic_3n), j = i - c * (3 * N);
                sU[c * US + j] = (i < tot) ? src[i] : T(0);
            }
        }
        if (tid < 18) { sMean[tid] = kk.mean[tid]; sInv[tid] = kk.inv_scale[tid]; }
        if (tid == 0) { s_prog[0] = 0; s_prog[1] = 0; s_best_c[6] = 0; }
    }
    // state (uniform loads), behind the controls' loads
    P0x = (T)sd_at(0); P0y = (T)sd_at(1); P0z = (T)sd_at(2);
    V0x = (T)sd_at(6); V0y = (T)sd_at(7); V0z = (T)sd_at(8);
    A0x = (T)sd_at(9); A0y = (T)sd_at(10); A0z = (T)sd_at(11);
    th0 = (T)sd_at(12); ga0 = (T)sd_at(13); thm0 = (T)sd_at(14); gam0 = (T)sd_at(15);
    const T P1x0 = (T)sd_at(3), P1y0 = (T)sd_at(4), P1z0 = (T)sd_at(5);
    __syncthreads();

    RV_STAMP(1);
    RV_STAMP_HW(2);
    RV_STAMP_SIMD(11);
    // ---- phase 2: exogenous feature rows of every node ------------------------------------
    // V_n (feature-frame velocity at node n) when it does not depend on (theta, gamma)
    auto vel = [&](int c, int node, T &vx, T &vy, T &vz) {
        if (node == 0) { vx = V0x; vy = V0y; vz = V0z; return; }
        const T *u = &sU[c * US + (node - 1) * 3];
        if (VT == ROVMPC_VT_TABLE) {
            const T *R = a.Rtab + (node - 1) * 9;                            // R @ v
            vx = R[0] * u[0] + R[1] * u[1] + R[2] * u[2];
            vy = R[3] * u[0] + R[4] * u[1] + R[5] * u[2];
            vz = R[6] * u[0] + R[7] * u[1] + R[8] * u[2];
        } else { vx = u[0]; vy = u[1]; vz = u[2]; }
    };
    // (compiled-in model: the gamma wave integrates gamma and fills its table meanwhile)
    if (gwave) {
        RV_STAMP_W(8);
        if (HANDOFF && a.from_ring) { ring_wait(a.seq_gamma); ga0 = ring_get(1); gam0 = ring_get(3); }
        // (long horizons, plain launches: a later round's workgroup finds the launch's gamma table published -- see gtab)
        bool have_gtab = false;
        if (!HANDOFF && !SAMPLE && MODEL == MODEL_BUILTIN && share_sines && a.gtab) {
            const int got = (tid & 63) == 0 ? (int)(ld_agent(a.gtab_tag) == (unsigned long long)a.epoch) : 0;
            have_gtab = __builtin_amdgcn_readfirstlane(got) != 0;
        }
        if (have_gtab) {
            for (int i = tid & 63; i < 8 * (N + 1); i += 64) sG[i] = ld_agent(&a.gtab[i]);
            if ((tid & 63) == 0) __hip_atomic_store(s_chain_done, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
        gamma_chain();
        if (HANDOFF) {
            if (a.from_ring || a.plant_feedback) {
                // sweeper: gamma_1 of this step is every candidate's gamma_1 -- the next step's gamma0, known already
                if ((int)blockIdx.x == a.sweeper && a.publish && a.plant_feedback) {
                    if ((tid & 63) == 0) {
                        const int nb = (int)((a.step + 1) & 3) * 4;
                        st_agent(&a.ring[nb + 1], (double)sG[5]);
                        st_agent(&a.ring[nb + 3], (double)ga0);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if ((tid & 63) == 0) st_agent(a.seq_gamma, (unsigned long long)(a.step + 1));
                }
            }
            if (MODEL == MODEL_BUILTIN && (tid & 63) == 0) sG[8 * N + 1] = ga0;           // phase 2b's gamma_0 (this wave alone knows it)
        }
        RV_STAMP_W(9);
        if (JGI) gamma_table_jit();
        else if (share_sines) { if ((tid & 63) == 0) __hip_atomic_store(s_chain_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
        else gamma_sines(tid & 63, 64, true);
        }
        RV_STAMP_W(10);
    }
    // One thread takes `p2m` CONSECUTIVE nodes of one candidate: the position of its first node is the n-term sum
    //   P_n = P_0 + sum_{j<n} (v_scale dt) U_j
    // in the reference's sequential order, every further node one more term of the same sum -- bit-identical to summing each
    // node from scratch, which is what round 2 did, one node per thread and round (an O(N^2) pass: a third of phase 2's
    // instructions at N = 20).  No scan, no extra barrier.
    auto node_item = [&](int n, int c, T Px, T Py, T Pz) {
        RV_PL(sP, 0, n, c) = Px; RV_PL(sP, 1, n, c) = Py; RV_PL(sP, 2, n, c) = Pz;
        const T rx = Px - P0x, ry = Py - P0y, rz = Pz - P0z;                 // simply.py:25
        {
            // rotation axes of the cable at this node: used by the velocity transform (phases 2b, 3) and
            // again by the augmented-catenary geometry (phase 4b)
            V3<T> kt, kg;
            theta_gamma_axes<T>({rx, ry, rz}, kt, kg);
            RV_PL(sA, 0, n, c) = kt.x; RV_PL(sA, 1, n, c) = kt.y;
            RV_PL(sA, 2, n, c) = kg.x; RV_PL(sA, 3, n, c) = kg.y; RV_PL(sA, 4, n, c) = kg.z;
        }
        if (MODEL == MODEL_BUILTIN) {
            if (VT != ROVMPC_VT_COMPOSE) {
                T Vx, Vy, Vz;
                vel(c, n, Vx, Vy, Vz);
                const T x3 = (Vx - sMean[3]) * sInv[3];                      // x3 is all the model reads
                RV_PX(0, n, c) = x3;
                if (!Trig<T>::bounded(m_abs(x3))) s_prog[0] = 1;             // a sine argument of the theta chain is huge (or NaN)
            }
            return;
        }
        const T nr = m_sqrt(rx * rx + ry * ry + rz * rz);
        const T inr = T(1) / (nr + T(1e-8));                                 // :26
        const T ux = rx * inr, uy = ry * inr, uz = rz * inr;
        if (fmap == ROVMPC_FEATURES_GEN3) {
            // features_dd (main_fun.py:839-864): plane p = scaled slot 4 + p =
            // [v_sway, v_surge, a_sway, a_surge, V(3), a(3)], velocities in m/s
            if (VT == ROVMPC_VT_COMPOSE) {       // V depends on (theta, gamma): rows are built beside the integration
                RV_PL(sA, 5, n, c) = ux; RV_PL(sA, 6, n, c) = uy; RV_PL(sA, 7, n, c) = uz;
                return;
            }
            // the neighbour node of the first difference: n - 1, or node 1 for n = 0 (np.gradient's edge rule, :846-847)
            const int m = n == 0 ? 1 : n - 1;
            const T *um = &sU[c * US];
            T Qx, Qy, Qz;                        // position of node m, bit-equal to its own sequential sum
            if (n == 0) { Qx = Px + kk.vs_h * um[0]; Qy = Py + kk.vs_h * um[1]; Qz = Pz + kk.vs_h * um[2]; }
            else {
                Qx = P1x0; Qy = P1y0; Qz = P1z0;
                for (int j = 0; j < m; ++j) { Qx = Qx + kk.vs_h * um[3 * j]; Qy = Qy + kk.vs_h * um[3 * j + 1]; Qz = Qz + kk.vs_h * um[3 * j + 2]; }
            }
            T Vx, Vy, Vz, Wx, Wy, Wz;
            vel(c, n, Vx, Vy, Vz); vel(c, m, Wx, Wy, Wz);
            const T qx = Qx - P0x, qy = Qy - P0y, qz = Qz - P0z;
            const T inq = m_div(T(1), m_sqrt(qx * qx + qy * qy + qz * qz) + T(1e-8));
            T sway_n, surge_n, sway_m, surge_m;
            dd_surge_sway<T>(kk.vs * Vx, kk.vs * Vy, kk.vs * Vz, ux, uy, uz, sway_n, surge_n);
            dd_surge_sway<T>(kk.vs * Wx, kk.vs * Wy, kk.vs * Wz, qx * inq, qy * inq, qz * inq, sway_m, surge_m);
            const T sgn = n == 0 ? T(-1) : T(1);
            const T a_sway = sgn * (sway_n - sway_m) * kk.inv_h, a_surge = sgn * (surge_n - surge_m) * kk.inv_h;
            T Ax, Ay, Az;
            if (n == 0) { Ax = A0x; Ay = A0y; Az = A0z; }
            else { Ax = (Vx - Wx) * kk.inv_h; Ay = (Vy - Wy) * kk.inv_h; Az = (Vz - Wz) * kk.inv_h; }
            const T row[10] = {sway_n, surge_n, a_sway, a_surge, kk.vs * Vx, kk.vs * Vy, kk.vs * Vz, kk.vs * Ax, kk.vs * Ay, kk.vs * Az};
#pragma unroll
            for (int p = 0; p < 10; ++p)
                if (uses(p)) RV_PX(p, n, c) = (row[p] - sMean[4 + p]) * sInv[4 + p];
            return;
        }
        const T tension = m_clip(nr, T(1e-5), T(10));                        // :27
        if (uses(0)) RV_PX(0, n, c) = (Px - sMean[0]) * sInv[0];
        if (uses(1)) RV_PX(1, n, c) = (Py - sMean[1]) * sInv[1];
        if (uses(2)) RV_PX(2, n, c) = (Pz - sMean[2]) * sInv[2];
        if (uses(9)) RV_PX(9, n, c) = (ux - sMean[9]) * sInv[9];
        if (uses(10)) RV_PX(10, n, c) = (uy - sMean[10]) * sInv[10];
        if (uses(11)) RV_PX(11, n, c) = (uz - sMean[11]) * sInv[11];
        if (uses(12)) RV_PX(12, n, c) = (tension - sMean[12]) * sInv[12];
        if (VT == ROVMPC_VT_COMPOSE) {
            RV_PL(sA, 5, n, c) = ux; RV_PL(sA, 6, n, c) = uy; RV_PL(sA, 7, n, c) = uz;
        } else {
            // velocity features do not depend on (theta, gamma): finish the row here
            T Vx, Vy, Vz, Wx, Wy, Wz;           // V_n and V_{n-1}
            vel(c, n, Vx, Vy, Vz);
            T Ax, Ay, Az;
            if (n == 0) { Ax = A0x; Ay = A0y; Az = A0z; }
            else { vel(c, n - 1, Wx, Wy, Wz); Ax = (Vx - Wx) * kk.inv_h; Ay = (Vy - Wy) * kk.inv_h; Az = (Vz - Wz) * kk.inv_h; }
            const T nv = m_sqrt(Vx * Vx + Vy * Vy + Vz * Vz) + T(1e-8);      // :30
            T ap = m_div(Vx * ux + Vy * uy + Vz * uz, nv);
            const int apslot = fmap == ROVMPC_FEATURES_GEN2 ? 16 : 13;
            if (fmap != ROVMPC_FEATURES_GEN2) ap = m_clip(ap, T(-1), T(1));              // :31 (generation 2 does not clip)
            if (uses(3)) RV_PX(3, n, c) = (Vx - sMean[3]) * sInv[3];
            if (uses(4)) RV_PX(4, n, c) = (Vy - sMean[4]) * sInv[4];
            if (uses(5)) RV_PX(5, n, c) = (Vz - sMean[5]) * sInv[5];
            if (uses(6)) RV_PX(6, n, c) = (Ax - sMean[6]) * sInv[6];
            if (uses(7)) RV_PX(7, n, c) = (Ay - sMean[7]) * sInv[7];
            if (uses(8)) RV_PX(8, n, c) = (Az - sMean[8]) * sInv[8];
            if (uses(13)) RV_PX(13, n, c) = (ap - sMean[apslot]) * sInv[apslot];
        }
    };
    if (MODEL == MODEL_BUILTIN) {
        const int p2m = ((N + 1) * CK + LNT - 1) / LNT;          // nodes per thread
        const int c = ltid & ckm, n0 = (ltid >> cks) * p2m;
        if (!(wideG && gwave) && !(a.debug & 4) && n0 <= N) {
            T Px = P1x0, Py = P1y0, Pz = P1z0;
            const T *u = &sU[c * US];
            for (int j = 0; j < n0; ++j) {
                Px = Px + kk.vs_h * u[3 * j]; Py = Py + kk.vs_h * u[3 * j + 1]; Pz = Pz + kk.vs_h * u[3 * j + 2];
            }
            const int n1 = min(n0 + p2m, N + 1);
            for (int n = n0; n < n1; ++n) {
                if (n > n0) { Px = Px + kk.vs_h * u[3 * n - 3]; Py = Py + kk.vs_h * u[3 * n - 2]; Pz = Pz + kk.vs_h * u[3 * n - 1]; }
                node_item(n, c, Px, Py, Pz);
            }
        }
    } else {
        // loaded models: one node per thread and round (their items carry the feature rows; measured: the consecutive-node
        // form above costs them 0.5 us at C2).  Full rounds take the far nodes in ascending order, the partial last round the
        // nodes next to the anchor (an item's cost grows with n).
        const int p2rem = (N + 1) % max(LNT >> cks, 1);
        for (int i = (wideG && gwave) ? (N + 1) * CK : ltid; i < ((a.debug & 4) ? 0 : (N + 1) * CK); i += LNT) {
            const int q = i >> cks, c = i & ckm;
            const int n = q < N + 1 - p2rem ? p2rem + q : q - (N + 1 - p2rem);
            T Px = P1x0, Py = P1y0, Pz = P1z0;
            const T *u = &sU[c * US];
            for (int j = 0; j < n; ++j) {
                Px = Px + kk.vs_h * u[3 * j]; Py = Py + kk.vs_h * u[3 * j + 1]; Pz = Pz + kk.vs_h * u[3 * j + 2];
            }
            node_item(n, c, Px, Py, Pz);
        }
    }
    // items taken through phase 4b early (compiled-in model, one theta wave, at least one pure geometry wave)
    int early = 0;
    if (MODEL == MODEL_BUILTIN && CK <= 16 && NT >= 64 + 64 + 64 && N * CK > 256) {
        const int r = (N * CK) % 256;
        if (r > 0 && r <= 64 && (r + CK - 1) / CK <= N / 2) early = r;
    }
    // (hiprtc-specialised models: one integrating wave, the others on the geometry; same early batch, taken by the first
    // geometry wave while the integration runs)
    if (MODEL == MODEL_JIT && CK <= 64 && NT >= 64 + 64 && N * CK > 256) {
        const int r = (N * CK) % 256;
        if (r > 0 && r <= 64 && (r + CK - 1) / CK <= N / 2) early = r;
    }
    // Long horizons (compiled-in model, one theta wave): when the join would leave MORE than one round of phase 4b, the
    // geometry waves chase the theta wave instead -- every pool thread owns the items tid, tid + pool, ..., takes them through
    // 4a at once and through 4b as soon as the theta wave has passed their node, so that after the last integration step only
    // the last nodes' batch is left (C3, N = 50: two rounds = 3.7 us after the join -> one batch).  With a single round left
    // (C2, the throughput geometries) the chase gains nothing -- that round runs on every SIMD after the join, a chased batch
    // on one -- and the extra waves slow the integrating one (measured in round 1): those keep the early batch only.
    const bool chase = LONGH && MODEL == MODEL_BUILTIN && CK <= 16 && wideB && N * CK - early > NT;
    // the integrating wave reports its progress only as far as somebody waits for it (the nodes of the early batch)
    const int prog_until = chase ? N : (early > 0 ? (early + CK - 1) / CK : 0);
    RV_STAMP(12);
    if (MODEL == MODEL_BUILTIN && share_sines) {
        while (__hip_atomic_load(s_chain_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(4);
        if (__hip_atomic_load(s_chain_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 1) gamma_sines(tid, NT, false);   // (2: the table was loaded whole)
    }
    __syncthreads();
    RV_STAMP(13);
    const bool gtab_loaded = MODEL == MODEL_BUILTIN && share_sines && *s_chain_done == 2;
    if (MODEL == MODEL_BUILTIN) {
        // the theta nodes somebody will wait for (early batch / chase) start as a marker: the waiting lane polls its own
        // node's slot -- the integrating wave publishes no progress word (eight instructions of its every step)
        for (int i = tid; i < prog_until * CK; i += NT) theta_slot_mark(&RV_PL(sY, 0, 1 + (i >> cks), i & ckm));
        if (share_sines && !gtab_loaded) gamma_G(tid, NT);           // (its reads are the shared sines; the theta chain reads it behind the next barrier)
        // ---- phase 2b: what hangs on gamma_n alone, for every (node, candidate): the gamma plane and the
        // first half of the velocity transform ------------------------------------------------------------
        // v_cat = R_theta(+theta_n) R_gamma(-gamma_n) u_n about the cable axes of node n (R @ v of
        // velocity_transform_batch.py:100-101, R composed from the augmentation angles).  gamma_n is in the
        // table, so w = R_gamma(-gamma_n) u_n is finished here; of R_theta w the model reads the x component
        // only, and with kt = (ktx, kty, 0)
        //   x3 = ((R_theta w).x - mean3) / scale3 = B' + C' cos(theta_n) + A' sin(theta_n),
        //   A = kty wz,  B = ktx (kt . w),  C = wx - B  (primes: scaled by 1/scale3, B' also shifted).
        for (int i = tid; i < ((a.debug & 4) ? 0 : (N + 1) * CK); i += NT) {
            const int n = i >> cks, c = i & ckm;
            RV_PL(sY, 1, n, c) = n == 0 ? (HANDOFF ? sG[8 * N + 1] : ga0) : sG[8 * (n - 1) + 5];
            if (VT != ROVMPC_VT_COMPOSE || n == N) continue;
            const T ktx = RV_PL(sA, 0, n, c), kty = RV_PL(sA, 1, n, c);
            const V3<T> kg = {RV_PL(sA, 2, n, c), RV_PL(sA, 3, n, c), RV_PL(sA, 4, n, c)};
            const T *u = &sU[c * US + n * 3];
            const V3<T> w = rodrigues_unit<T>({u[0], u[1], u[2]}, kg, -sG[8 * n], sG[8 * n + 1]);
            const T B = ktx * (ktx * w.x + kty * w.y);
            const T Ap = (kty * w.z) * sInv[3], Bp = (B - sMean[3]) * sInv[3], Cp = (w.x - B) * sInv[3];
            RV_PL(sA, 5, n, c) = Ap; RV_PL(sA, 6, n, c) = Bp; RV_PL(sA, 7, n, c) = Cp;
            // |x3| <= |A'| + |B'| + |C'| whatever theta is: decides here whether the theta chain may use the
            // sine without the large-argument branch
            if (!Trig<T>::bounded((m_abs(Ap) + m_abs(Bp)) + m_abs(Cp))) s_prog[0] = 1;
        }
        __syncthreads();
        // workgroup 0 publishes its finished gamma table for the later rounds of the grid (write-through stores by one wave,
        // drained, then the epoch tag: G16 form R1 with sc1 on both sides, as the epilogue's trajectories)
        if (!HANDOFF && !SAMPLE && share_sines && a.gtab && !gtab_loaded && blockIdx.x == 0 && gwave) {
            for (int i = tid & 63; i < 8 * (N + 1); i += 64) st_agent(&a.gtab[i], sG[i]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((tid & 63) == 0) st_agent(a.gtab_tag, (unsigned long long)a.epoch);
        }
    }
    // Loaded model with a candidate-invariant gamma path, composed velocity transform: as above, the first half of the transform
    // w = R_gamma(-gamma_n) u_n hangs on the table alone and is finished here for every (node, candidate), in parallel, instead
    // of on the integrating lane's chain (planes 5..7 carry w: they hold the unit vector only for a model that reads
    // angle_proj, which keeps the rotation on the chain)
    const bool jgi_w = JGI && VT == ROVMPC_VT_COMPOSE && (used & 0x21f8u) != 0 && !uses(13);
    if (jgi_w) {
        for (int i = tid; i < ((a.debug & 4) ? 0 : N * CK); i += NT) {
            const int n = i >> cks, c = i & ckm;
            const V3<T> kg = {RV_PL(sA, 2, n, c), RV_PL(sA, 3, n, c), RV_PL(sA, 4, n, c)};
            const T *u = &sU[c * US + n * 3];
            const V3<T> w = rodrigues_unit<T>({u[0], u[1], u[2]}, kg, -sG[GROW * n], sG[GROW * n + 1]);
            RV_PL(sA, 5, n, c) = w.x; RV_PL(sA, 6, n, c) = w.y; RV_PL(sA, 7, n, c) = w.z;
        }
        __syncthreads();
    }

    RV_STAMP(3);
    // ---- phase 3: closed-loop integration of (theta, gamma), with the state-independent half of
    // the per-node geometry (phase 4a) running beside it on the workgroup's other waves --------

    // phase 4a, item (n, c): node n+1 of candidate c.  Nothing here depends on (theta, gamma):
    // catenary parameter + tension of the straight geometry (main_fun.py:292-293, 303-305),
    // tautness and control terms of the cost.
    // work list of one thread: optionally one own item first, then items begin + first + k stride
    auto geometry_a = [&](int own_item, int first, int stride, int begin) {
        const int total = (a.debug & 2) ? 0 : N * CK;
        bool pending_own = own_item >= 0;
        for (int i = pending_own ? own_item : begin + first; i < total;) {
            const int n = i >> cks, c = i & ckm;
            const T rx = RV_PL(sP, 0, n + 1, c) - P0x, ry = RV_PL(sP, 1, n + 1, c) - P0y,
                    rz = RV_PL(sP, 2, n + 1, c) - P0z;
            const T *u = &sU[c * US + n * 3];
            const T l = m_sqrtq(rx * rx + ry * ry);                              // main_fun.py:292
            const T dH = kk.up * rz;                                              // :293
            const T d = m_sqrtq(rx * rx + ry * ry + rz * rz);
            const CatRoot<T> cr = solve_catenary_root<T>(l, dH, kk.L, kk.c_lo, kk.c_hi);   // :303
            const T Tn = cable_tension<T>(l, cr, kk.w_per_len);                   // :304-305
            sW[n * CK + c] = (cr.C == cr.C) ? cr.u : T(-1);                      // warm start of phase 4b's solve of this node:
            sW[(N + n) * CK + c] = cr.e;                                          // the root and its exp
            const T e0 = u[0] - kk.Uref[0], e1 = u[1] - kk.Uref[1], e2 = u[2] - kk.Uref[2];
            const T taut = m_max(T(0), d - kk.rhoL);
            sC[n * CK + c] = kk.w_u * (e0 * e0 + e1 * e1 + e2 * e2) + kk.w_T * Tn + kk.w_taut * (taut * taut);
            if (pending_own) { pending_own = false; i = begin + first; } else i += stride;
        }
    };

    // phase 4b, item (n, c): node n+1 of candidate c, needs (theta, gamma)_{n+1}: the theta-rotated
    // end point, its catenary, the lowest z of the augmented shape (main_fun.py:38-111,
    // fully_augmented_catenary.py:21-22) and the rest of the cost.
    const Trig<T> trig4(false);
    auto geometry_b_item = [&](int n, int c) {
        const T rx = RV_PL(sP, 0, n + 1, c) - P0x, ry = RV_PL(sP, 1, n + 1, c) - P0y,
                rz = RV_PL(sP, 2, n + 1, c) - P0z;
        const T th = RV_PL(sY, 0, n + 1, c), ga = RV_PL(sY, 1, n + 1, c);
        const V3<T> kt = {RV_PL(sA, 0, n + 1, c), RV_PL(sA, 1, n + 1, c), T(0)};
        const V3<T> kg = {RV_PL(sA, 2, n + 1, c), RV_PL(sA, 3, n + 1, c), RV_PL(sA, 4, n + 1, c)};
        T st, ct, sg, cg;
        trig4.sincos(th, &st, &ct);
        if (MODEL == MODEL_BUILTIN) {
            // sincos(gamma_{n+1}) is candidate-invariant: the gamma wave left it in its table (node N: row N, slots 2 / 3)
            const int gi = n + 1 < N ? 8 * (n + 1) : 8 * N + 2;
            sg = sG[gi]; cg = sG[gi + 1];
        } else if (JGI) {
            sg = sG[GROW * (n + 1)]; cg = sG[GROW * (n + 1) + 1];
        } else {
            trig4.sincos(ga, &sg, &cg);
        }
        const AugShape<T> sh = augmented_prepare<T>({rx, ry, rz}, kt, kg, st, ct, sg, cg, kk.up);
        const CatRoot<T> cr = solve_catenary_root_warm<T>(sh.lp, sh.dHp, kk.L, kk.c_lo, kk.c_hi, sW[n * CK + c], sW[(N + n) * CK + c]);   // Catenary(A, B')
        const T zl = P0z + augmented_finish<T>(sh, cr, kk.L, a.M, kk.inv_Mm1, kk.up);
        const T eth = th - kk.theta_ref, ega = ga - kk.gamma_ref;
        const T flo = m_max(T(0), kk.up * (kk.z_floor - zl));
        sC[n * CK + c] = kk.w_theta * (eth * eth) + kk.w_gamma * (ega * ega) + sC[n * CK + c] + kk.w_floor * (flo * flo);
    };
    if (MODEL == MODEL_BUILTIN) {
        // saved_models/equations_dtheta_dt.csv complexity 13:
        //   ((((sin(x17) - sin(x3)) - x16) - x3) * 0.048152514)      -- no dependence on the stage state
        // saved_models/equations_dgamma_dt.csv complexity 3:  (x15 - x17)
        // Of the 18 slots only x3, x15, x16, x17 are read.  Three structural facts of these rows
        // shape the phase:
        //  * the gamma path is candidate-invariant and control-free: it was finished beside phase 2a
        //    (gamma_chain / gamma_sines), and with it everything that hangs on gamma_n alone -- the x17
        //    sines and the first half of the velocity transform (phase 2b);
        //  * dtheta/dt does not read the stage state, so the four RK4 slopes of a step are known once
        //    (theta_n-1, theta_n) and x3 at both ends of the step are.  With x3m = (x3a + x3b) / 2 and
        //    the interpolated delay slot p16m = (s16a + s16b) / 2 the RK4 sum collapses to
        //      k1 + 2 k2 + 2 k3 + k4 = KT [ G_n - (sin x3a + 4 sin x3m + sin x3b) - 3 (s16a + s16b) - 3 (x3a + x3b) ],
        //    G_n = sin x17a + 4 sin x17m + sin x17e from the table (HOLD: 6 s16a; Euler: k1 alone);
        //  * what is left on the sequential chain per step: x3b = B' + C' cos(theta_n) + A' sin(theta_n), two sines
        //    (ONE polynomial evaluation over the wave: each candidate owns a quad, lane = 4 c + role,
        //    roles exchange with DPP quad_perm moves), the sum above, and sincos(theta_n+1) by angle addition.
        const int nint = nintB;
        const bool hold = a.prev_mode == ROVMPC_PREV_HOLD;
        const bool euler = a.integrator == ROVMPC_EULER;
        const int nsteps = (a.debug & 1) ? 0 : N;

        auto theta_path = [&]() {
            const int lane = tid & 63, c16 = lane >> 2, role = lane & 3;
            const int cc = (tid >> 6) * 16 + c16;               // candidate of this lane
            const bool live = cc < CK;
            const int c = live ? cc : CK - 1;                   // clamp reads of padding lanes
            const Trig<T> trig(true);
            const T m3 = sMean[3], i3 = sInv[3], m16 = sMean[16], i16 = sInv[16];
            const T KT = T(0.048152514);
            const T hKT = euler ? kk.h * KT : (kk.h / T(6)) * KT;
            const T K16a = T(3) * i16, K16b = T(-6) * m16 * i16;
            if (HANDOFF && a.wait_theta) {
                ring_wait(a.seq_theta);
                if (a.from_ring) { th0 = ring_get(0); thm0 = ring_get(2); }
            }
            // Loop-carried values live in two-element arrays indexed by the step's parity p (a literal in each of the two
            // inlined copies of the step): TH[p] = theta of the previous node, TH[p ^ 1] = theta of this node; X3[p], SX[p] =
            // x3 and its sine at the step's start; O*[p] = the step's operands, fetched one step ahead into O*[p ^ 1].  A step
            // writes its results over the values that die with it, so the hand-over to the next step is a renaming, not a
            // dozen register moves (the optimiser refuses to unroll this loop itself: it holds wave-level votes).
            T TH[2] = {thm0, th0};
            if (live && role == 0) RV_PL(sY, 0, 0, c) = th0;
            T X3[2], SX[2];
            X3[0] = (V0x - m3) * i3; X3[1] = T(0);
            SX[0] = trig.sin(X3[0]); SX[1] = T(0);
            // RK4 with the interpolated delay slot, arranged for the chain's DEPTH (one wave issues a dependent instruction
            // every ~9.5 cycles, an independent one every ~5: tools/micro/fma_latency.hip):
            //  * the lane's sine argument -- x3b on the end lane, (x3a + x3b) / 2 on the midpoint lane -- is ONE FMA,
            //    hf * x3b + HA with hf = 1 | 1/2 and HA = 0 | x3a / 2 (the same value as the sum halved: halving is exact);
            //  * every node's share of the sum, w = -sin x3 - 3 x3 - K16b / 2, is formed once and carried to the next step;
            //    what waits for the step's sines is  S = (base - 4 sin x3m) + w_end  with base = G_n + w_start - K16a (theta_{n-1}
            //    + theta_n) ready long before them.
            const T hf = (role & 1) ? T(0.5) : T(1), hq = (role & 1) ? T(0.5) : T(0);
            const T mK16bh = T(-0.5) * K16b;
            T HA[2] = {hq * X3[0], T(0)};
            T W[2] = {m_fma(T(-3), X3[0], mK16bh) - SX[0], T(0)};
            // operands of step n, fetched one step ahead (nothing in the loop waits on another wave)
            T OA[2] = {T(0), T(0)}, OB[2] = {T(0), T(0)}, OC[2] = {T(0), T(0)}, GG[2] = {T(0), T(0)};
            auto fetch = [&](int n, int q) {
                if (VT == ROVMPC_VT_COMPOSE) { OA[q] = RV_PL(sA, 5, n, c); OB[q] = RV_PL(sA, 6, n, c); OC[q] = RV_PL(sA, 7, n, c); }
                else OB[q] = RV_PX(0, n + 1, c);
                GG[q] = sG[8 * n + 2];
            };
            if (nsteps > 0) fetch(0, 0);
            const bool bounded = s_prog[0] == 0 && Trig<T>::bounded(m_abs(X3[0]));   // every sine argument of the loop is below the fast-path limit
            // sincos(theta_n) for the velocity transform.  theta moves by |d| ~ 1e-4 per step, so
            // after a full evaluation at step 0 (and every 16th step, or whenever a lane's |d|
            // reaches 2^-7) the pair is advanced by the angle-addition formulas with the odd/even
            // Taylor polynomials of d to d^5 / d^6 (truncation < 4e-19 there).
            T st = T(0), ct = T(1);
            if (VT == ROVMPC_VT_COMPOSE) trig.sincos(th0, &st, &ct);
            // One loop for the common mode (RK4, interpolated delay slot, bounded sine arguments) with its flags
            // as literals, one for everything else with run-time flags.
            auto one_step = [&](auto FAST, auto PAR, int n) {
                constexpr int p = PAR.value ? 1 : 0, q = p ^ 1;
                const bool eul = FAST.value ? false : euler, hld = FAST.value ? false : hold, bnd = FAST.value ? true : bounded;
                const T thm = TH[p], th = TH[q], x3a = X3[p], sinXa = SX[p];
                const T Gn = GG[p];
                const T x3b = VT == ROVMPC_VT_COMPOSE ? (OB[p] + OC[p] * ct) + OA[p] * st : OB[p];
                // (the operands of a step past the horizon are never used; with the composed transform their rows exist --
                // the planes and the table have N + 1 of them -- so the last step fetches like the others, without a guard)
                if (VT == ROVMPC_VT_COMPOSE || n + 1 < nsteps) fetch(n + 1, q);
                const T sarg = m_fma(hf, x3b, HA[p]);                      // x3b | the feature midpoint (x3a + x3b) / 2 (:62)
                const T s2 = bnd ? trig.sin_bounded(sarg) : trig.sin(sarg);
                T s2r[4];
                quad4(s2, s2r);
                const T sinXb = s2r[0], sinXm = s2r[1];
                T S;
                if (eul || hld) {
                    // delay slot x16 at the start of the step (np.roll semantics, simply.py:35-38)
                    const T s16a = (thm - m16) * i16;
                    if (eul) S = ((Gn - sinXa) - s16a) - x3a;                       // main_fun.py:761
                    else S = ((Gn - ((sinXa + sinXb) + T(4) * sinXm)) - T(6) * s16a) - T(3) * (x3a + x3b);   // :66, delay slot held
                } else {
                    // :66 -- 3 (s16a + s16b) = K16a (theta_{n-1} + theta_n) + K16b
                    const T base = m_fma(-K16a, thm + th, Gn + W[p]);
                    const T wb = m_fma(T(-3), x3b, mK16bh) - sinXb;
                    S = m_fma(T(-4), sinXm, base) + wb;
                    W[q] = wb;
                }
                HA[q] = hq * x3b;
                const T thn = th + hKT * S;
                if (live && role == 0) RV_PL(sY, 0, n + 1, c) = thn;    // (whoever waits for this node polls the slot: theta_slot_wait)
                if (VT == ROVMPC_VT_COMPOSE) {
                    // per lane: angle addition; the full evaluation at the anchors and for a lane whose own step is large -- a
                    // candidate's arithmetic never depends on its neighbours in the wave.  The vote on the large lanes comes
                    // AFTER the addition: behind the compare it would stall the chain in every step (a wave issues a dependent
                    // instruction every ~9.5 cycles), and a branch in the middle of the step keeps the scheduler from filling
                    // the sine's dependent chain with the step's independent work.
                    const T dlt = thn - th;
                    const bool big = !(m_abs(dlt) < T(0.015625));
                    // |d| < 2^-6: the first neglected terms, d^9 / 362880 and d^8 / 40320, are below 1e-19 (absolute, of
                    // values of size one): sixteen steps between two anchors add up to less than 2e-18.  (The bound was
                    // 2^-7 with one term less: on the synthetic workload one wave-step in eight then held a lane beyond it.)
                    const T d2 = dlt * dlt;
                    const T sd = dlt * (T(1) - d2 * T(1.0 / 6) * (T(1) - d2 * T(1.0 / 20) * (T(1) - d2 * T(1.0 / 42))));
                    const T cd = T(1) - d2 * T(0.5) * (T(1) - d2 * T(1.0 / 12) * (T(1) - d2 * T(1.0 / 30)));
                    const T sn = st * cd + ct * sd;
                    ct = ct * cd - st * sd;
                    st = sn;
                    const bool anchor = PAR.value && ((n + 1) & 15) == 0;        // (anchors fall on odd steps)
                    if (__builtin_expect(anchor || __any(big), 0)) {              // rare (the hint moves the block out of the loop's instruction stream)
                        T fs, fc;
                        trig.sincos(thn, &fs, &fc);
                        if (anchor || big) { st = fs; ct = fc; }
                    }
                }
                TH[p] = thn;                                   // over theta_{n-1}, dead from here
                X3[q] = x3b; SX[q] = sinXb;
            };
            auto run = [&](auto FAST) {
                int n = 0;
                for (; n + 1 < nsteps; n += 2) { one_step(FAST, BoolC<false>{}, n); one_step(FAST, BoolC<true>{}, n + 1); }
                if (n < nsteps) one_step(FAST, BoolC<false>{}, n);
            };
            if (!euler && !hold && bounded) run(BoolC<true>{}); else run(BoolC<false>{});
            if (tid == 0) __hip_atomic_store(&s_prog[1], N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };

        // Dispatch (every routine has ONE call site: the kernel is run once through per launch, so
        // its size is instruction-cache misses).  Wide workgroups: theta waves | everybody else on the
        // state-independent geometry.  Narrow ones (a single wave): the theta path, then the geometry.
        const bool wide = wideB;
        if (tid < nint) {
            __builtin_amdgcn_s_setprio(3);           // the workgroup's critical path
            theta_path();
            __builtin_amdgcn_s_setprio(0);
        }
        // Phase 4a pool = every thread that does not integrate theta (all threads when narrow).
        // Early batch: N*CK items on a workgroup of N*CK threads are one wave more than the CU has
        // SIMDs (320 items = 5 waves on 4 SIMDs: one SIMD would issue two waves' worth of phase 4b
        // after the join).  The first `early` items -- the nodes the theta wave finishes first --
        // are therefore taken through 4a AND 4b by an otherwise idle wave while the integration is
        // still running; the join then leaves a multiple of 256.
        const int j = wide ? tid - nint - 64 : -1;
        const bool own = !chase && j >= 0 && j < early && !(a.debug & 2);
        if (!wide || tid >= nint) geometry_a(own ? j : -1, wide ? tid - nint : tid, wide ? NT - nint : NT, chase ? 0 : early);
        // phase 4b before the join: the early batch's item, or -- chasing -- all of this pool thread's items in node order
        int b0 = 0, b1 = 0, bs = 1;
        if (chase) { if (tid >= nint && !(a.debug & 2)) { b0 = tid - nint; b1 = N * CK; bs = NT - nint; } }
        else if (own) { b0 = j; b1 = j + 1; }
        for (int i = b0; i < b1; i += bs) {
            const int n = i >> cks;
            // until the node's theta is there -- or the chain is through (a theta that IS the marker's bit pattern, a NaN
            // payload handed in with the inputs, is then simply read)
            while (theta_slot_marked(&RV_PL(sY, 0, n + 1, i & ckm)) &&
                   __hip_atomic_load(&s_prog[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < N) __builtin_amdgcn_s_sleep(4);
            geometry_b_item(n, i & ckm);
        }
        if (chase) early = N * CK;                    // nothing is left for the round after the join
    } else {
        // bytecode model: CK lanes of wave 0 integrate; the other waves take phase 4a
        const int nint = 64;
        auto integrate = [&]() {
            const Trig<T> trigj(false);
            if (HANDOFF && a.wait_theta) {        // (models without the gamma shortcut take all four from the end-of-step record)
                ring_wait(a.seq_theta);
                if (a.from_ring) { th0 = ring_get(0); ga0 = ring_get(1); thm0 = ring_get(2); gam0 = ring_get(3); }
            }
            if (tid >= CK) return;
            const int c = tid;
            const int nsteps = (a.debug & 1) ? 0 : N;
            T th = th0, ga = ga0, thm = thm0, gam = gam0;
            RV_PL(sY, 0, 0, c) = th; RV_PL(sY, 1, 0, c) = ga;
            const T m14 = sMean[14], i14 = sInv[14], m15 = sMean[15], i15 = sInv[15];
            const T m16 = sMean[16], i16 = sInv[16], m17 = sMean[17], i17 = sInv[17];
            const bool hold = a.prev_mode == ROVMPC_PREV_HOLD;
            const bool euler = a.integrator == ROVMPC_EULER;
            const T hstep = kk.h, inv_hstep = kk.inv_h;
            const bool gen2 = fmap == ROVMPC_FEATURES_GEN2;
            const T hh = T(0.5) * hstep, h6 = hstep / T(6);
            // generic path: full 18-slot feature row per stage, bytecode interpreter
            T Vx = V0x, Vy = V0y, Vz = V0z;
            auto store_vslots = [&](int node, T vx, T vy, T vz, T ax, T ay, T az) {
                if (uses(13)) {
                    const T ux = RV_PL(sA, 5, node, c), uy = RV_PL(sA, 6, node, c), uz = RV_PL(sA, 7, node, c);
                    const T nv = m_sqrt(vx * vx + vy * vy + vz * vz) + T(1e-8);
                    T ap = (vx * ux + vy * uy + vz * uz) / nv;
                    const int apslot = gen2 ? 16 : 13;
                    if (!gen2) ap = m_clip(ap, T(-1), T(1));
                    RV_PX(13, node, c) = (ap - sMean[apslot]) * sInv[apslot];
                }
                if (uses(3)) RV_PX(3, node, c) = (vx - sMean[3]) * sInv[3];
                if (uses(4)) RV_PX(4, node, c) = (vy - sMean[4]) * sInv[4];
                if (uses(5)) RV_PX(5, node, c) = (vz - sMean[5]) * sInv[5];
                if (uses(6)) RV_PX(6, node, c) = (ax - sMean[6]) * sInv[6];
                if (uses(7)) RV_PX(7, node, c) = (ay - sMean[7]) * sInv[7];
                if (uses(8)) RV_PX(8, node, c) = (az - sMean[8]) * sInv[8];
            };
            const bool vel_used = (used & 0x21f8u) != 0;     // planes 3..8, 13
            if (VT == ROVMPC_VT_COMPOSE && vel_used) store_vslots(0, Vx, Vy, Vz, A0x, A0y, A0z);
            T *feat = sF + c;                       // [slot][lane]
            T *stack = sF + 18 * CK + c;
            for (int n = 0; n < nsteps; ++n) {
                if (VT == ROVMPC_VT_COMPOSE && vel_used) {
                    const V3<T> kt = {RV_PL(sA, 0, n, c), RV_PL(sA, 1, n, c), T(0)};
                    const V3<T> kg = {RV_PL(sA, 2, n, c), RV_PL(sA, 3, n, c), RV_PL(sA, 4, n, c)};
                    T st, ct, sg, cg;
                    m_sincos(th, &st, &ct); m_sincos(ga, &sg, &cg);
                    const T *u = &sU[c * US + n * 3];
                    V3<T> v = rodrigues_unit<T>({u[0], u[1], u[2]}, kg, -sg, cg);
                    v = rodrigues_flat<T>(v, kt.x, kt.y, st, ct);
                    store_vslots(n + 1, v.x, v.y, v.z, (v.x - Vx) * inv_hstep, (v.y - Vy) * inv_hstep, (v.z - Vz) * inv_hstep);
                    Vx = v.x; Vy = v.y; Vz = v.z;
                }
                const T s16a = (thm - m16) * i16, s16b = (th - m16) * i16;
                const T s17a = (gam - m17) * i17, s17b = (ga - m17) * i17;
                // one stage: f(features(y, t_n + cfrac h)); cfrac2 = 2 cfrac in {0, 1, 2}
                auto stage = [&](T yth, T yga, int cfrac2, T &dth, T &dga) {
                    T p16, p17;
                    if (hold || cfrac2 == 0) { p16 = s16a; p17 = s17a; }
                    else if (cfrac2 == 2) { p16 = s16b; p17 = s17b; }
                    else { p16 = (s16a + s16b) / T(2); p17 = (s17a + s17b) / T(2); }
                    if (MODEL == MODEL_JIT) {
                        // features in registers; loads of slots the expressions never read are dead
                        T x[18];
#pragma unroll
                        for (int s = 0; s < NEXO; ++s) {
                            if (cfrac2 == 0) x[s] = RV_PX(s, n, c);
                            else if (cfrac2 == 2) x[s] = RV_PX(s, n + 1, c);
                            else x[s] = (RV_PX(s, n, c) + RV_PX(s, n + 1, c)) / T(2);   // :62
                        }
                        if (gen2) {
                            // simulate_rk4_theta_gamma.py:40: [.., unit_rel, theta, gamma, cos(theta), sin(gamma), angle_proj]
                            x[16] = x[13];                     // plane 13 carries angle_proj
                            x[12] = (yth - sMean[12]) * sInv[12]; x[13] = (yga - sMean[13]) * sInv[13];
                            x[14] = (m_cos(yth) - m14) * i14; x[15] = (m_sin(yga) - m15) * i15;
                            x[17] = T(0);
                        } else {
                            x[14] = (yth - m14) * i14; x[15] = (yga - m15) * i15; x[16] = p16; x[17] = p17;
                        }
                        T e[ROVMPC_JIT_NSUB > 0 ? ROVMPC_JIT_NSUB : 1];
                        jit_exo<T>(x, e, trigj);
                        dth = jit_f_theta<T>(x, e, (const T *)nullptr, trigj);
                        dga = jit_f_gamma<T>(x, e, trigj);
                        return;
                    }
                    for (int s = 0; s < NEXO; ++s) {
                        if (!uses(s)) continue;
                        T v;
                        if (cfrac2 == 0) v = RV_PX(s, n, c);
                        else if (cfrac2 == 2) v = RV_PX(s, n + 1, c);
                        else v = (RV_PX(s, n, c) + RV_PX(s, n + 1, c)) / T(2);      // :62
                        feat[s * CK] = v;
                    }
                    if (gen2) {
                        feat[16 * CK] = feat[13 * CK];
                        feat[12 * CK] = (yth - sMean[12]) * sInv[12]; feat[13 * CK] = (yga - sMean[13]) * sInv[13];
                        feat[14 * CK] = (m_cos(yth) - m14) * i14; feat[15 * CK] = (m_sin(yga) - m15) * i15;
                        feat[17 * CK] = T(0);
                    } else {
                        feat[14 * CK] = (yth - m14) * i14; feat[15 * CK] = (yga - m15) * i15;
                        feat[16 * CK] = p16; feat[17 * CK] = p17;
                    }
                    dth = interp_eval<T>(a.code_th, a.n_th, a.consts, feat, CK, stack, CK);
                    dga = interp_eval<T>(a.code_ga, a.n_ga, a.consts, feat, CK, stack, CK);
                };
                T k1t, k1g;
                stage(th, ga, 0, k1t, k1g);
                T thn, gan;
                if (euler) {
                    thn = th + k1t * hstep;                                     // main_fun.py:761
                    gan = ga + k1g * hstep;
                } else {
                    T k2t, k2g, k3t, k3g, k4t, k4g;
                    stage(th + hh * k1t, ga + hh * k1g, 1, k2t, k2g);
                    stage(th + hh * k2t, ga + hh * k2g, 1, k3t, k3g);
                    stage(th + hstep * k3t, ga + hstep * k3g, 2, k4t, k4g);
                    thn = th + h6 * (k1t + T(2) * k2t + T(2) * k3t + k4t);    // :66
                    gan = ga + h6 * (k1g + T(2) * k2g + T(2) * k3g + k4g);
                }
                thm = th; gam = ga; th = thn; ga = gan;
                RV_PL(sY, 0, n + 1, c) = th; RV_PL(sY, 1, n + 1, c) = ga;
            }
        };
        // Second-order generation (features_dd): y = (theta, gamma, dtheta, dgamma), y' = (dtheta, dgamma, f_theta(x), f_gamma(x)).
        // Slots 0..3 of the row are the stage state, planes 0..9 the exogenous slots 4..13 (midpoint between nodes at
        // the half stages, simulate_rk4_theta_gamma.py:62).  integrator EULER = the reference's explicit double Euler
        // (test_cluster.py:113-129).  State slots 14/15 carry (dtheta, dgamma) at node 0.
        auto integrate_dd = [&]() {
            const Trig<T> trigj(false);
            const T vs_reg = kk.vs;       // a register for the loop (the constants live in global memory: one load per trip otherwise)
            if (HANDOFF && a.wait_theta) {        // (models without the gamma shortcut take all four from the end-of-step record)
                ring_wait(a.seq_theta);
                if (a.from_ring) { th0 = ring_get(0); ga0 = ring_get(1); thm0 = ring_get(2); gam0 = ring_get(3); }
            }
            if (tid >= CK) return;
            const int c = tid;
            const int nsteps = (a.debug & 1) ? 0 : N;
            T y0 = th0, y1 = ga0, y2 = thm0, y3 = gam0;
            RV_PL(sY, 0, 0, c) = y0; RV_PL(sY, 1, 0, c) = y1;
            const bool euler = a.integrator == ROVMPC_EULER;
            const T hstep = kk.h, inv_hstep = kk.inv_h, hh = T(0.5) * kk.h, h6 = kk.h / T(6);
            T Vx = V0x, Vy = V0y, Vz = V0z, sway_p = T(0), surge_p = T(0);
            if (VT == ROVMPC_VT_COMPOSE)
                dd_surge_sway<T>(vs_reg * Vx, vs_reg * Vy, vs_reg * Vz, RV_PL(sA, 5, 0, c), RV_PL(sA, 6, 0, c), RV_PL(sA, 7, 0, c), sway_p, surge_p);
            auto store_row = [&](int node, T sway, T surge, T a_sway, T a_surge, T vx, T vy, T vz, T ax, T ay, T az) {
                const T row[10] = {sway, surge, a_sway, a_surge, vs_reg * vx, vs_reg * vy, vs_reg * vz, vs_reg * ax, vs_reg * ay, vs_reg * az};
#pragma unroll
                for (int p = 0; p < 10; ++p)
                    if (uses(p)) RV_PX(p, node, c) = (row[p] - sMean[4 + p]) * sInv[4 + p];
            };
            T *feat = sF + c;
            T *stack = sF + 18 * CK + c;
            for (int n = 0; n < nsteps; ++n) {
                if (VT == ROVMPC_VT_COMPOSE && (used & 0x3ffu)) {
                    const V3<T> kt = {RV_PL(sA, 0, n, c), RV_PL(sA, 1, n, c), T(0)};
                    const V3<T> kg = {RV_PL(sA, 2, n, c), RV_PL(sA, 3, n, c), RV_PL(sA, 4, n, c)};
                    T st, ct, sg, cg;
                    m_sincos(y0, &st, &ct); m_sincos(y1, &sg, &cg);
                    const T *u = &sU[c * US + n * 3];
                    V3<T> v = rodrigues_unit<T>({u[0], u[1], u[2]}, kg, -sg, cg);
                    v = rodrigues_flat<T>(v, kt.x, kt.y, st, ct);
                    T sway_n, surge_n;
                    dd_surge_sway<T>(vs_reg * v.x, vs_reg * v.y, vs_reg * v.z, RV_PL(sA, 5, n + 1, c), RV_PL(sA, 6, n + 1, c), RV_PL(sA, 7, n + 1, c), sway_n, surge_n);
                    const T a_sway = (sway_n - sway_p) * inv_hstep, a_surge = (surge_n - surge_p) * inv_hstep;
                    if (n == 0) store_row(0, sway_p, surge_p, a_sway, a_surge, Vx, Vy, Vz, A0x, A0y, A0z);
                    store_row(n + 1, sway_n, surge_n, a_sway, a_surge, v.x, v.y, v.z,
                              (v.x - Vx) * inv_hstep, (v.y - Vy) * inv_hstep, (v.z - Vz) * inv_hstep);
                    Vx = v.x; Vy = v.y; Vz = v.z; sway_p = sway_n; surge_p = surge_n;
                }
                auto stage = [&](T s0, T s1, T s2, T s3, int cfrac2, T &ddth, T &ddga) {
                    if (MODEL == MODEL_JIT) {
                        T x[18];
#pragma unroll
                        for (int p = 0; p < 10; ++p) {
                            if (cfrac2 == 0) x[4 + p] = RV_PX(p, n, c);
                            else if (cfrac2 == 2) x[4 + p] = RV_PX(p, n + 1, c);
                            else x[4 + p] = (RV_PX(p, n, c) + RV_PX(p, n + 1, c)) / T(2);
                        }
                        x[0] = (s0 - sMean[0]) * sInv[0]; x[1] = (s1 - sMean[1]) * sInv[1];
                        x[2] = (s2 - sMean[2]) * sInv[2]; x[3] = (s3 - sMean[3]) * sInv[3];
                        x[14] = x[15] = x[16] = x[17] = T(0);
                        T e[ROVMPC_JIT_NSUB > 0 ? ROVMPC_JIT_NSUB : 1];
                        jit_exo<T>(x, e, trigj);
                        ddth = jit_f_theta<T>(x, e, (const T *)nullptr, trigj);
                        ddga = jit_f_gamma<T>(x, e, trigj);
                        return;
                    }
                    for (int p = 0; p < 10; ++p) {
                        if (!uses(p)) continue;
                        T v;
                        if (cfrac2 == 0) v = RV_PX(p, n, c);
                        else if (cfrac2 == 2) v = RV_PX(p, n + 1, c);
                        else v = (RV_PX(p, n, c) + RV_PX(p, n + 1, c)) / T(2);
                        feat[(4 + p) * CK] = v;
                    }
                    feat[0] = (s0 - sMean[0]) * sInv[0]; feat[CK] = (s1 - sMean[1]) * sInv[1];
                    feat[2 * CK] = (s2 - sMean[2]) * sInv[2]; feat[3 * CK] = (s3 - sMean[3]) * sInv[3];
                    ddth = interp_eval<T>(a.code_th, a.n_th, a.consts, feat, CK, stack, CK);
                    ddga = interp_eval<T>(a.code_ga, a.n_ga, a.consts, feat, CK, stack, CK);
                };
                T a1t, a1g;
                stage(y0, y1, y2, y3, 0, a1t, a1g);
                if (euler) {
                    const T n0 = y0 + y2 * hstep, n1 = y1 + y3 * hstep;            // test_cluster.py:125-129
                    y2 = y2 + a1t * hstep; y3 = y3 + a1g * hstep;                    // :113-117
                    y0 = n0; y1 = n1;
                } else {
                    // k_i = (rate_i, acc_i); rates at the stage states
                    const T r1t = y2, r1g = y3;
                    const T r2t = y2 + hh * a1t, r2g = y3 + hh * a1g;
                    T a2t, a2g, a3t, a3g, a4t, a4g;
                    stage(y0 + hh * r1t, y1 + hh * r1g, r2t, r2g, 1, a2t, a2g);
                    const T r3t = y2 + hh * a2t, r3g = y3 + hh * a2g;
                    stage(y0 + hh * r2t, y1 + hh * r2g, r3t, r3g, 1, a3t, a3g);
                    const T r4t = y2 + hstep * a3t, r4g = y3 + hstep * a3g;
                    stage(y0 + hstep * r3t, y1 + hstep * r3g, r4t, r4g, 2, a4t, a4g);
                    y0 = y0 + h6 * (r1t + T(2) * r2t + T(2) * r3t + r4t);
                    y1 = y1 + h6 * (r1g + T(2) * r2g + T(2) * r3g + r4g);
                    y2 = y2 + h6 * (a1t + T(2) * a2t + T(2) * a3t + a4t);
                    y3 = y3 + h6 * (a1g + T(2) * a2g + T(2) * a3g + a4g);
                }
                RV_PL(sY, 0, n + 1, c) = y0; RV_PL(sY, 1, n + 1, c) = y1;
            }
        };

        // ---- hiprtc-specialised models: the same integration with the per-step instruction chain cut down -------------
        // (measured at C2: generation 2 rows 58.6 -> see profiles/r02_other_configs.jsonl).  What changes against the
        // loop above, none of it in the arithmetic of a stage:
        //  * the exogenous rows of the two step ends live in registers (xa, xb); the velocity-dependent slots, which
        //    only this lane ever reads, never go through LDS (no store -> load round trip on the chain, and no LDS
        //    planes for them);
        //  * the operands of step n + 1 (axes, control, unit vector, state-independent rows) are fetched during step n;
        //  * sincos(theta_n), sincos(gamma_n) -- velocity transform, generation-2 slots cos(theta) / sin(gamma) at the
        //    stage states -- advance by the angle-addition formulas from the previous evaluation (odd / even Taylor
        //    polynomials of the increment to d^5 / d^6, truncation < 4e-19 for |d| < 2^-7), re-anchored by a full
        //    evaluation every 16 steps and for any lane whose increment is larger (wave-uniform choice, per-lane
        //    result: a candidate's arithmetic never depends on its neighbours).
        auto add_angle = [](T s0, T c0, T d, T &s, T &c) {
            const T d2 = d * d;
            const T sd = d * (T(1) - d2 * T(1.0 / 6) * (T(1) - d2 * T(1.0 / 20)));
            const T cd = T(1) - d2 * T(0.5) * (T(1) - d2 * T(1.0 / 12) * (T(1) - d2 * T(1.0 / 30)));
            s = s0 * cd + c0 * sd; c = c0 * cd - s0 * sd;
        };
        // the pair (theta, gamma) at once: one wave vote for both
        auto sincos_near2 = [&](T x, T xanchor, T sa, T ca, T &s, T &c, T y, T yanchor, T sb, T cb, T &s2, T &c2) {
            const T d = x - xanchor, e = y - yanchor;
            const bool bigx = !(m_abs(d) < T(0.0078125)), bigy = !(m_abs(e) < T(0.0078125));
            T rs, rc, qs, qc;
            add_angle(sa, ca, d, rs, rc);
            add_angle(sb, cb, e, qs, qc);
            if (__builtin_expect(__any(bigx | bigy), 0)) {
                T fs, fc;
                m_sincos(x, &fs, &fc);
                if (bigx) { rs = fs; rc = fc; }
                m_sincos(y, &fs, &fc);
                if (bigy) { qs = fs; qc = fc; }
            }
            s = rs; c = rc; s2 = qs; c2 = qc;
        };
        constexpr unsigned VELMASK = 0x21f8u;       // planes 3..8 and 13: velocity, acceleration, angle_proj
        auto integrate_jit = [&]() {
            // the sines of the generated expressions run on the chain: their coefficients are pinned in registers once (a 64-bit
            // literal costs two moves per use)
            const Trig<T> trigj(ROVMPC_JIT_PIN_TRIG != 0);
            if (HANDOFF && a.wait_theta) {        // (models without the gamma shortcut take all four from the end-of-step record)
                ring_wait(a.seq_theta);
                if (a.from_ring) { th0 = ring_get(0); ga0 = ring_get(1); thm0 = ring_get(2); gam0 = ring_get(3); }
            }
            if (tid >= CK) return;
            const int c = tid;
            const int nsteps = (a.debug & 1) ? 0 : N;
            T th = th0, ga = ga0, thm = thm0, gam = gam0;
            RV_PL(sY, 0, 0, c) = th; RV_PL(sY, 1, 0, c) = ga;
            const T m14 = sMean[14], i14 = sInv[14], m15 = sMean[15], i15 = sInv[15];
            const T m16 = sMean[16], i16 = sInv[16], m17 = sMean[17], i17 = sInv[17];
            const bool hold_rt = a.prev_mode == ROVMPC_PREV_HOLD;
            const bool euler_rt = a.integrator == ROVMPC_EULER;
            const T hstep = kk.h, inv_hstep = kk.inv_h, hh = T(0.5) * hstep, h6 = hstep / T(6);
            const bool gen2 = fmap == ROVMPC_FEATURES_GEN2;
            const bool compose_rows = VT == ROVMPC_VT_COMPOSE && (used & VELMASK) != 0;
            auto in_lds = [&](int sl) { return uses(sl) && !(VT == ROVMPC_VT_COMPOSE && ((VELMASK >> sl) & 1u)); };
            const int apslot = gen2 ? 16 : 13;
            T xa[NEXO], xb[NEXO];
#pragma unroll
            for (int sl = 0; sl < NEXO; ++sl) { xa[sl] = xb[sl] = T(0); if (in_lds(sl)) xa[sl] = RV_PX(sl, 0, c); }
            // velocity-dependent slots of one node from (v, a, unit vector): simply.py:29-31, scaled.  The scaling pairs of the
            // slots in use sit in registers: read in the step they are an LDS round trip on the chain, every step (the
            // progress word's store keeps the compiler from hoisting them)
            T vmean[9], vinv[9];
#pragma unroll
            for (int sl = 3; sl < 9; ++sl) { vmean[sl] = vinv[sl] = T(0); if (uses(sl)) { vmean[sl] = sMean[sl]; vinv[sl] = sInv[sl]; } }
            const T apmean = uses(13) ? sMean[apslot] : T(0), apinv = uses(13) ? sInv[apslot] : T(0);
            const T g2m12 = gen2 ? sMean[12] : T(0), g2i12 = gen2 ? sInv[12] : T(0), g2m13 = gen2 ? sMean[13] : T(0), g2i13 = gen2 ? sInv[13] : T(0);
            auto vel_slots = [&](T *x, T vx, T vy, T vz, T ax, T ay, T az, T ux, T uy, T uz) {
                if (uses(13)) {
                    const T nv = m_sqrtq(vx * vx + vy * vy + vz * vz) + T(1e-8);
                    T ap = (vx * ux + vy * uy + vz * uz) * fast_rcp(nv);      // nv >= 1e-8 (or NaN / inf, which stay that)
                    if (!gen2) ap = m_clip(ap, T(-1), T(1));
                    x[13] = (ap - apmean) * apinv;
                }
                if (uses(3)) x[3] = (vx - vmean[3]) * vinv[3];
                if (uses(4)) x[4] = (vy - vmean[4]) * vinv[4];
                if (uses(5)) x[5] = (vz - vmean[5]) * vinv[5];
                if (uses(6)) x[6] = (ax - vmean[6]) * vinv[6];
                if (uses(7)) x[7] = (ay - vmean[7]) * vinv[7];
                if (uses(8)) x[8] = (az - vmean[8]) * vinv[8];
            };
            T Vx = V0x, Vy = V0y, Vz = V0z;
            if (compose_rows)
                vel_slots(xa, Vx, Vy, Vz, A0x, A0y, A0z, RV_PL(sA, 5, 0, c), RV_PL(sA, 6, 0, c), RV_PL(sA, 7, 0, c));
            // operands of a step: rotation axes and control of node n, unit vector of node n + 1
            // (ROVMPC_JIT_GI: + what the step reads of the gamma wave's table -- gamma_{n+1}, its sincos, gamma's stage states, the
            // x17-only subexpressions on the three rows -- fetched a step ahead like the rest: a table read inside the step is an
            // LDS round trip on the chain)
            constexpr int NGT = (ROVMPC_JIT_GI && ROVMPC_JIT_NGSUB > 0) ? ROVMPC_JIT_NGSUB : 1;
            struct Ops { T ktx, kty, kgx, kgy, kgz, u0, u1, u2, ux, uy, uz; T gnext, sgn, cgn, gst2, gst3, gst4, gs[3 * NGT]; };
            Ops opA = {}, opB = {};
            auto fetch_ops = [&](int n, Ops &o) {
                if (ROVMPC_JIT_GI) {
                    o.gnext = sG[GROW * n + 5]; o.sgn = sG[GROW * (n + 1)]; o.cgn = sG[GROW * (n + 1) + 1];
                    if (!ROVMPC_JIT_TS) { o.gst2 = sG[GROW * n + 2]; o.gst3 = sG[GROW * n + 3]; o.gst4 = sG[GROW * n + 4]; }
                    if (ROVMPC_JIT_NGSUB > 0) {
#pragma unroll
                        for (int k = 0; k < 3 * NGT; ++k) o.gs[k] = sG[GROW * n + 8 + k];
                    }
                }
                if (!compose_rows) return;
                o.ktx = RV_PL(sA, 0, n, c); o.kty = RV_PL(sA, 1, n, c);
                if (jgi_w) {                       // (u0..u2 carry w = R_gamma(-gamma_n) u_n, finished before the integration)
                    o.u0 = RV_PL(sA, 5, n, c); o.u1 = RV_PL(sA, 6, n, c); o.u2 = RV_PL(sA, 7, n, c);
                } else {
                    o.kgx = RV_PL(sA, 2, n, c); o.kgy = RV_PL(sA, 3, n, c); o.kgz = RV_PL(sA, 4, n, c);
                    const T *u = &sU[c * US + n * 3];
                    o.u0 = u[0]; o.u1 = u[1]; o.u2 = u[2];
                }
                if (uses(13)) { o.ux = RV_PL(sA, 5, n + 1, c); o.uy = RV_PL(sA, 6, n + 1, c); o.uz = RV_PL(sA, 7, n + 1, c); }
            };
            auto fetch_rows = [&](int node, T *x) {
#pragma unroll
                for (int sl = 0; sl < NEXO; ++sl) if (in_lds(sl)) x[sl] = RV_PX(sl, node, c);
            };
            if (nsteps > 0) { fetch_ops(0, opA); fetch_rows(1, xb); }
            // sincos of the node state (velocity transform; anchor of the generation-2 slots)
            const bool need_trig = compose_rows || gen2;
            T st = T(0), ct = T(1), sg = T(0), cg = T(1);
            if (need_trig) { m_sincos(th, &st, &ct); m_sincos(ga, &sg, &cg); }
            // One step: A = row of node n (start), B = row of node n + 1 (end; its velocity slots are built here), o = the
            // step's operands, onext = where the next step's are fetched.  Once the third stage is through, A is dead and
            // takes the row of node n + 2: the next step runs with the roles swapped, so the hand-over is a renaming.
            // the model's stage-invariant subexpressions (jit_exo) on the start row, the end row and their midpoint: the end row
            // of a step is the start row of the next, stages two and three share the midpoint -- two evaluations per step
            constexpr int NSUB = ROVMPC_JIT_NSUB > 0 ? ROVMPC_JIT_NSUB : 1;
            T ea[NSUB], eb[NSUB], em[NSUB];
            // the slots of a stage that no state enters, from row A (cfrac2 = 0), the midpoint (1), row B (2)
            auto fill_exo = [&](T *x, const T *A, const T *B, int cfrac2) {
#pragma unroll
                for (int sl = 0; sl < NEXO; ++sl)
                    x[sl] = cfrac2 == 0 ? A[sl] : (cfrac2 == 2 ? B[sl] : (A[sl] + B[sl]) / T(2));   // :62
                if (gen2) x[16] = x[13];        // simulate_rk4_theta_gamma.py:40: plane 13 carries angle_proj, slot 16 there
            };
            auto exo_subs = [&](const T *A, const T *B, int cfrac2, T *e) {
                if (ROVMPC_JIT_NSUB == 0) return;
                T x[18];
#pragma unroll
                for (int sl = 0; sl < 18; ++sl) x[sl] = T(0);
                fill_exo(x, A, B, cfrac2);
                jit_exo<T>(x, e, trigj);
            };
            exo_subs(xa, xb, 0, ea);
            // ROVMPC_JIT_TS: the end row's slope of a step is the next step's start slope when the delay slots are interpolated
            // (HOLD keeps the previous node's value through the step: its end row is not the next start row) and RK4 evaluates it
            T kA_carry = T(0);
            auto one_step = [&](int n, T *A, T *B, T *eA, T *eB, const Ops &o, Ops &onext, auto FAST) {
                // FAST: the common mode (RK4, interpolated delay slots) with its flags as literals
                const bool euler = FAST.value ? false : euler_rt, hold = FAST.value ? false : hold_rt;
                const bool carry_ok = ROVMPC_JIT_TS && !hold && !euler;
                if (n + 1 < nsteps) fetch_ops(n + 1, onext);
                if (compose_rows) {
                    const V3<T> kt = {o.ktx, o.kty, T(0)}, kg = {o.kgx, o.kgy, o.kgz};
                    V3<T> v = {o.u0, o.u1, o.u2};
                    if (!jgi_w) v = rodrigues_unit<T>(v, kg, -sg, cg);
                    v = rodrigues_flat<T>(v, kt.x, kt.y, st, ct);
                    vel_slots(B, v.x, v.y, v.z, (v.x - Vx) * inv_hstep, (v.y - Vy) * inv_hstep, (v.z - Vz) * inv_hstep, o.ux, o.uy, o.uz);
                    Vx = v.x; Vy = v.y; Vz = v.z;
                }
                exo_subs(A, B, 2, eB);
                if (!euler) exo_subs(A, B, 1, em);
                const T s16a = (thm - m16) * i16, s16b = (th - m16) * i16;
                const T s17a = (gam - m17) * i17, s17b = (ga - m17) * i17;
                // the x17-only subexpressions of dtheta/dt on this step's three rows, from the gamma wave's table
                constexpr int NG = (ROVMPC_JIT_GI && ROVMPC_JIT_NGSUB > 0) ? ROVMPC_JIT_NGSUB : 1;
                T gsa[NG], gsm[NG], gsb[NG];
                if (ROVMPC_JIT_GI && ROVMPC_JIT_NGSUB > 0) {
#pragma unroll
                    for (int k = 0; k < NG; ++k) { gsa[k] = o.gs[3 * k]; gsm[k] = o.gs[3 * k + 1]; gsb[k] = o.gs[3 * k + 2]; }
                }
                // one evaluation of the pair at stage state (yth, yga) on the start (cfrac2 = 0), midpoint (1) or end (2) row
                auto stage = [&](T yth, T yga, int cfrac2, T &dth, T &dga, auto WANT_TH, auto WANT_GA) {
                    T p16, p17;
                    if (hold || cfrac2 == 0) { p16 = s16a; p17 = s17a; }
                    else if (cfrac2 == 2) { p16 = s16b; p17 = s17b; }
                    else { p16 = (s16a + s16b) / T(2); p17 = (s17a + s17b) / T(2); }
                    T x[18];
                    fill_exo(x, A, B, cfrac2);
                    const T *e = cfrac2 == 0 ? eA : (cfrac2 == 2 ? eB : em);
                    const T *g = cfrac2 == 0 ? gsa : (cfrac2 == 2 ? gsb : gsm);
                    if (gen2) {
                        // simulate_rk4_theta_gamma.py:40: [.., unit_rel, theta, gamma, cos(theta), sin(gamma), angle_proj]
                        x[12] = (yth - g2m12) * g2i12; x[13] = (yga - g2m13) * g2i13;
                        T s_t = st, c_t = ct, s_g = sg, c_g = cg;        // first stage: the node state itself
                        if (cfrac2 != 0) sincos_near2(yth, th, st, ct, s_t, c_t, yga, ga, sg, cg, s_g, c_g);
                        x[14] = (c_t - m14) * i14; x[15] = (s_g - m15) * i15;
                        x[17] = T(0);
                    } else {
                        x[14] = (yth - m14) * i14; x[15] = (yga - m15) * i15; x[16] = p16; x[17] = p17;
                    }
                    if (WANT_TH.value) dth = jit_f_theta<T>(x, e, g, trigj);
                    if (WANT_GA.value) dga = jit_f_gamma<T>(x, e, trigj);
                };
                constexpr BoolC<true> YES{};
                constexpr BoolC<false> NO{};
                T thn, gan;
                if (ROVMPC_JIT_TS) {
                    // dtheta/dt reads no stage state: its slopes are f on the start row, the midpoint row (twice) and the end row
                    T dummy = T(0);
                    T kat = kA_carry, kmt = T(0), kbt = T(0);
                    if (!carry_ok || n == 0) stage(th, ga, 0, kat, dummy, YES, NO);
                    if (!euler) stage(th, ga, 1, kmt, dummy, YES, NO);
                    if (!euler || carry_ok) stage(th, ga, 2, kbt, dummy, YES, NO);
                    kA_carry = kbt;
                    thn = euler ? th + kat * hstep : th + h6 * (kat + T(2) * kmt + T(2) * kmt + kbt);    // :66 with k2 = k3
                    if (ROVMPC_JIT_GI) {
                        gan = o.gnext;                                           // the gamma wave's chain
                    } else {
                        // gamma's own stages, with theta's stage states known beforehand
                        T k1g, k2g, k3g, k4g;
                        stage(th, ga, 0, dummy, k1g, NO, YES);
                        if (euler) {
                            gan = ga + k1g * hstep;
                        } else {
                            stage(th + hh * kat, ga + hh * k1g, 1, dummy, k2g, NO, YES);
                            stage(th + hh * kmt, ga + hh * k2g, 1, dummy, k3g, NO, YES);
                            stage(th + hstep * kmt, ga + hstep * k3g, 2, dummy, k4g, NO, YES);
                            gan = ga + h6 * (k1g + T(2) * k2g + T(2) * k3g + k4g);
                        }
                    }
                    if (n + 1 < nsteps) fetch_rows(n + 2, A);                    // A is dead from here on
                } else if (ROVMPC_JIT_GI) {
                    // gamma's path and stage states come from the gamma wave's table; dtheta/dt runs its stages on them
                    T dummy = T(0), k1t, k2t, k3t, k4t;
                    stage(th, ga, 0, k1t, dummy, YES, NO);
                    if (euler) {
                        thn = th + k1t * hstep;
                    } else {
                        stage(th + hh * k1t, o.gst2, 1, k2t, dummy, YES, NO);
                        stage(th + hh * k2t, o.gst3, 1, k3t, dummy, YES, NO);
                        stage(th + hstep * k3t, o.gst4, 2, k4t, dummy, YES, NO);
                        thn = th + h6 * (k1t + T(2) * k2t + T(2) * k3t + k4t);
                    }
                    gan = o.gnext;
                    if (n + 1 < nsteps) fetch_rows(n + 2, A);
                } else {
                T k1t, k1g;
                stage(th, ga, 0, k1t, k1g, YES, YES);
                if (euler) {
                    thn = th + k1t * hstep;                                     // main_fun.py:761
                    gan = ga + k1g * hstep;
                    if (n + 1 < nsteps) fetch_rows(n + 2, A);
                } else {
                    T k2t, k2g, k3t, k3g, k4t, k4g;
                    stage(th + hh * k1t, ga + hh * k1g, 1, k2t, k2g, YES, YES);
                    stage(th + hh * k2t, ga + hh * k2g, 1, k3t, k3g, YES, YES);
                    if (n + 1 < nsteps) fetch_rows(n + 2, A);                  // A is dead from here on
                    stage(th + hstep * k3t, ga + hstep * k3g, 2, k4t, k4g, YES, YES);
                    thn = th + h6 * (k1t + T(2) * k2t + T(2) * k3t + k4t);    // :66
                    gan = ga + h6 * (k1g + T(2) * k2g + T(2) * k3g + k4g);
                }
                }
                if (ROVMPC_JIT_GI) {
                    // sincos(gamma_{n+1}) is in the table; theta alone advances by angle addition
                    if (need_trig && n + 1 < nsteps) {
                        sg = o.sgn; cg = o.cgn;
                        // (as in the compiled-in chain: the addition always, anchors and large steps as ONE rare block that the
                        // hint moves out of the loop's instruction stream)
                        const T d = thn - th;
                        const bool big = !(m_abs(d) < T(0.0078125)), anchor = ((n + 1) & 15) == 0;
                        T rs, rc;
                        add_angle(st, ct, d, rs, rc);
                        if (__builtin_expect(anchor || __any(big), 0)) { T fs, fc; m_sincos(thn, &fs, &fc); if (anchor || big) { rs = fs; rc = fc; } }
                        st = rs; ct = rc;
                    }
                } else
                if (need_trig && n + 1 < nsteps) {
                    if (((n + 1) & 15) == 0) { m_sincos(thn, &st, &ct); m_sincos(gan, &sg, &cg); }
                    else sincos_near2(thn, th, st, ct, st, ct, gan, ga, sg, cg, sg, cg);
                }
                thm = th; gam = ga; th = thn; ga = gan;
                RV_PL(sY, 0, n + 1, c) = th; RV_PL(sY, 1, n + 1, c) = ga;
                // progress word for the early phase-4b batch (one wave's DS operations complete in order)
                if (n < prog_until && tid == 0) __hip_atomic_store(&s_prog[1], n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            auto run = [&](auto FAST) {
                int n = 0;
                for (; n + 1 < nsteps; n += 2) { one_step(n, xa, xb, ea, eb, opA, opB, FAST); one_step(n + 1, xb, xa, eb, ea, opB, opA, FAST); }
                if (n < nsteps) one_step(n, xa, xb, ea, eb, opA, opB, FAST);
            };
            if (!euler_rt && !hold_rt) run(BoolC<true>{}); else run(BoolC<false>{});
        };
        // second-order generation, same scheme: every exogenous slot of features_dd hangs on the velocity, so under
        // VT_COMPOSE no row goes through LDS at all
        auto integrate_dd_jit = [&]() {
            const Trig<T> trigj(false);
            const T vs_reg = kk.vs;       // a register for the loop (the constants live in global memory: one load per trip otherwise)
            if (HANDOFF && a.wait_theta) {        // (models without the gamma shortcut take all four from the end-of-step record)
                ring_wait(a.seq_theta);
                if (a.from_ring) { th0 = ring_get(0); ga0 = ring_get(1); thm0 = ring_get(2); gam0 = ring_get(3); }
            }
            if (tid >= CK) return;
            const int c = tid;
            const int nsteps = (a.debug & 1) ? 0 : N;
            T y0 = th0, y1 = ga0, y2 = thm0, y3 = gam0;
            RV_PL(sY, 0, 0, c) = y0; RV_PL(sY, 1, 0, c) = y1;
            const bool euler_rt = a.integrator == ROVMPC_EULER;
            const T hstep = kk.h, inv_hstep = kk.inv_h, hh = T(0.5) * kk.h, h6 = kk.h / T(6);
            const bool compose_rows = VT == ROVMPC_VT_COMPOSE && (used & 0x3ffu) != 0;
            auto in_lds = [&](int p) { return uses(p) && VT != ROVMPC_VT_COMPOSE; };
            T xa[10], xb[10];
#pragma unroll
            for (int p = 0; p < 10; ++p) { xa[p] = xb[p] = T(0); if (in_lds(p)) xa[p] = RV_PX(p, 0, c); }
            // the scaling pairs of the state slots and of the exogenous slots in use, in registers (read in the step they are an
            // LDS round trip on the chain: the progress word's store keeps the compiler from hoisting them)
            T dmean[14], dinv[14];
#pragma unroll
            for (int p = 0; p < 14; ++p) { dmean[p] = dinv[p] = T(0); if (p < 4 || uses(p - 4)) { dmean[p] = sMean[p]; dinv[p] = sInv[p]; } }
            auto row = [&](T *x, T sway, T surge, T a_sway, T a_surge, T vx, T vy, T vz, T ax, T ay, T az) {
                const T r[10] = {sway, surge, a_sway, a_surge, vs_reg * vx, vs_reg * vy, vs_reg * vz, vs_reg * ax, vs_reg * ay, vs_reg * az};
#pragma unroll
                for (int p = 0; p < 10; ++p) if (uses(p)) x[p] = (r[p] - dmean[4 + p]) * dinv[4 + p];
            };
            T Vx = V0x, Vy = V0y, Vz = V0z, sway_p = T(0), surge_p = T(0);
            if (compose_rows)
                dd_surge_sway<T>(vs_reg * Vx, vs_reg * Vy, vs_reg * Vz, RV_PL(sA, 5, 0, c), RV_PL(sA, 6, 0, c), RV_PL(sA, 7, 0, c), sway_p, surge_p);
            struct Ops { T ktx, kty, kgx, kgy, kgz, u0, u1, u2, ux, uy, uz; };
            Ops opA = {}, opB = {};
            auto fetch_ops = [&](int n, Ops &o) {
                if (!compose_rows) return;
                o.ktx = RV_PL(sA, 0, n, c); o.kty = RV_PL(sA, 1, n, c);
                o.kgx = RV_PL(sA, 2, n, c); o.kgy = RV_PL(sA, 3, n, c); o.kgz = RV_PL(sA, 4, n, c);
                const T *u = &sU[c * US + n * 3];
                o.u0 = u[0]; o.u1 = u[1]; o.u2 = u[2];
                o.ux = RV_PL(sA, 5, n + 1, c); o.uy = RV_PL(sA, 6, n + 1, c); o.uz = RV_PL(sA, 7, n + 1, c);
            };
            auto fetch_rows = [&](int node, T *x) {
#pragma unroll
                for (int p = 0; p < 10; ++p) if (in_lds(p)) x[p] = RV_PX(p, node, c);
            };
            if (nsteps > 0) { fetch_ops(0, opA); fetch_rows(1, xb); }
            T st = T(0), ct = T(1), sg = T(0), cg = T(1);
            if (compose_rows) { m_sincos(y0, &st, &ct); m_sincos(y1, &sg, &cg); }
            constexpr int NSUB = ROVMPC_JIT_NSUB > 0 ? ROVMPC_JIT_NSUB : 1;
            T ea[NSUB], eb[NSUB], em[NSUB];           // as in integrate_jit
            auto fill_exo = [&](T *x, const T *A, const T *B, int cfrac2) {
#pragma unroll
                for (int p = 0; p < 10; ++p)
                    x[4 + p] = cfrac2 == 0 ? A[p] : (cfrac2 == 2 ? B[p] : (A[p] + B[p]) / T(2));
                x[14] = x[15] = x[16] = x[17] = T(0);
            };
            auto exo_subs = [&](const T *A, const T *B, int cfrac2, T *e) {
                if (ROVMPC_JIT_NSUB == 0) return;
                T x[18];
                x[0] = x[1] = x[2] = x[3] = T(0);
                fill_exo(x, A, B, cfrac2);
                jit_exo<T>(x, e, trigj);
            };
            bool ea_ready = false;
            auto one_step = [&](int n, T *A, T *B, T *eA, T *eB, const Ops &o, Ops &onext, auto FAST) {
                const bool euler = FAST.value ? false : euler_rt;          // FAST: RK4 with the flag as a literal
                if (n + 1 < nsteps) fetch_ops(n + 1, onext);
                if (compose_rows) {
                    const V3<T> kt = {o.ktx, o.kty, T(0)}, kg = {o.kgx, o.kgy, o.kgz};
                    V3<T> v = rodrigues_unit<T>({o.u0, o.u1, o.u2}, kg, -sg, cg);
                    v = rodrigues_flat<T>(v, kt.x, kt.y, st, ct);
                    T sway_n, surge_n;
                    dd_surge_sway<T>(vs_reg * v.x, vs_reg * v.y, vs_reg * v.z, o.ux, o.uy, o.uz, sway_n, surge_n);
                    const T a_sway = (sway_n - sway_p) * inv_hstep, a_surge = (surge_n - surge_p) * inv_hstep;
                    if (n == 0) row(A, sway_p, surge_p, a_sway, a_surge, Vx, Vy, Vz, A0x, A0y, A0z);   // np.gradient's edge rule
                    row(B, sway_n, surge_n, a_sway, a_surge, v.x, v.y, v.z,
                        (v.x - Vx) * inv_hstep, (v.y - Vy) * inv_hstep, (v.z - Vz) * inv_hstep);
                    Vx = v.x; Vy = v.y; Vz = v.z; sway_p = sway_n; surge_p = surge_n;
                }
                // (row A of the first step is complete only here: np.gradient's edge rule needs the first step's end row)
                if (!ea_ready) { exo_subs(A, B, 0, eA); ea_ready = true; }
                exo_subs(A, B, 2, eB);
                if (!euler) exo_subs(A, B, 1, em);
                auto stage = [&](T s0, T s1, T s2, T s3, int cfrac2, T &ddth, T &ddga) {
                    T x[18];
                    fill_exo(x, A, B, cfrac2);
                    const T *e = cfrac2 == 0 ? eA : (cfrac2 == 2 ? eB : em);
                    x[0] = (s0 - dmean[0]) * dinv[0]; x[1] = (s1 - dmean[1]) * dinv[1];
                    x[2] = (s2 - dmean[2]) * dinv[2]; x[3] = (s3 - dmean[3]) * dinv[3];
                    ddth = jit_f_theta<T>(x, e, (const T *)nullptr, trigj);
                    ddga = jit_f_gamma<T>(x, e, trigj);
                };
                T a1t, a1g;
                stage(y0, y1, y2, y3, 0, a1t, a1g);
                T n0, n1;
                if (euler) {
                    n0 = y0 + y2 * hstep; n1 = y1 + y3 * hstep;                      // test_cluster.py:125-129
                    y2 = y2 + a1t * hstep; y3 = y3 + a1g * hstep;                    // :113-117
                    if (n + 1 < nsteps) fetch_rows(n + 2, A);
                } else {
                    const T r1t = y2, r1g = y3;
                    const T r2t = y2 + hh * a1t, r2g = y3 + hh * a1g;
                    T a2t, a2g, a3t, a3g, a4t, a4g;
                    stage(y0 + hh * r1t, y1 + hh * r1g, r2t, r2g, 1, a2t, a2g);
                    const T r3t = y2 + hh * a2t, r3g = y3 + hh * a2g;
                    stage(y0 + hh * r2t, y1 + hh * r2g, r3t, r3g, 1, a3t, a3g);
                    if (n + 1 < nsteps) fetch_rows(n + 2, A);                       // A is dead from here on
                    const T r4t = y2 + hstep * a3t, r4g = y3 + hstep * a3g;
                    stage(y0 + hstep * r3t, y1 + hstep * r3g, r4t, r4g, 2, a4t, a4g);
                    n0 = y0 + h6 * (r1t + T(2) * r2t + T(2) * r3t + r4t);
                    n1 = y1 + h6 * (r1g + T(2) * r2g + T(2) * r3g + r4g);
                    y2 = y2 + h6 * (a1t + T(2) * a2t + T(2) * a3t + a4t);
                    y3 = y3 + h6 * (a1g + T(2) * a2g + T(2) * a3g + a4g);
                }
                if (compose_rows && n + 1 < nsteps) {
                    if (((n + 1) & 15) == 0) { m_sincos(n0, &st, &ct); m_sincos(n1, &sg, &cg); }
                    else sincos_near2(n0, y0, st, ct, st, ct, n1, y1, sg, cg, sg, cg);
                }
                y0 = n0; y1 = n1;
                RV_PL(sY, 0, n + 1, c) = y0; RV_PL(sY, 1, n + 1, c) = y1;
                if (n < prog_until && tid == 0) __hip_atomic_store(&s_prog[1], n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            auto run = [&](auto FAST) {
                int n = 0;
                for (; n + 1 < nsteps; n += 2) { one_step(n, xa, xb, ea, eb, opA, opB, FAST); one_step(n + 1, xb, xa, eb, ea, opB, opA, FAST); }
                if (n < nsteps) one_step(n, xa, xb, ea, eb, opA, opB, FAST);
            };
            if (!euler_rt) run(BoolC<true>{}); else run(BoolC<false>{});
        };
        const bool wide = NT > nint;
        if (tid < nint) {
            if (MODEL == MODEL_JIT) { if (fmap == ROVMPC_FEATURES_GEN3) integrate_dd_jit(); else integrate_jit(); }
            else { if (fmap == ROVMPC_FEATURES_GEN3) integrate_dd(); else integrate(); }
            if (tid == 0) __hip_atomic_store(&s_prog[1], N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (also when no step ran)
        }
        const int j = (wide && MODEL == MODEL_JIT) ? tid - nint : -1;
        const bool own = j >= 0 && j < early && !(a.debug & 2);
        if (!wide || tid >= nint) geometry_a(own ? j : -1, wide ? tid - nint : tid, wide ? NT - nint : NT, wide ? early : 0);
        if (own) {
            const int n = j >> cks;
            while (__hip_atomic_load(&s_prog[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < n + 1) __builtin_amdgcn_s_sleep(8);
            geometry_b_item(n, j & ckm);
        }
    }
    __syncthreads();

    RV_STAMP(4);
    // ---- phase 4b ---------------------------------------------------------------------------
    for (int i = early + tid; i < ((a.debug & 2) ? 0 : N * CK); i += NT) geometry_b_item(i >> cks, i & ckm);
    __syncthreads();

    RV_STAMP(5);
    // ---- phase 5: J_k, block arg-min, outputs ---------------------------------------------
    // per-problem views of the outputs (computed here, not at the top: nothing before this point needs them live)
    T *Jb = a.J + (size_t)prob * K;
    T *trajb = a.traj_all ? a.traj_all + (size_t)prob * K * (N + 1) * 2 : nullptr;
    double *blk_trajb = a.blk_traj + (size_t)prob * a.nblocks * (N + 1) * 2;
    unsigned long long *granb = a.granules + (size_t)prob * GRAN * a.nblocks;
    double *resultb = a.result ? a.result + (size_t)prob * (5 + 2 * (N + 1)) : nullptr;
    const bool fast_tail = resultb && !a.traj_all && N + 1 <= 64;
    double &s_best_J = *reinterpret_cast<double *>(reinterpret_cast<char *>(smem) + 16);   // header bytes 16..23
    if (tid < 64) {
        const int c = tid;
        double Jd = __builtin_inf();
        long long kk = 0x7fffffffffffffffLL;
        if (c < nvalid) {
            // sequential sum (the reference order).  The terms are fetched ten at a time with clamped addresses
            // (no branch between the loads), so the LDS latency is paid once per chunk, not once per addition;
            // terms past the horizon enter as +0.
            T J = T(0);
            for (int n0 = 0; n0 < N; n0 += 10) {
                T v[10];
#pragma unroll
                for (int e = 0; e < 10; ++e) v[e] = sC[min(n0 + e, N - 1) * CK + c];
#pragma unroll
                for (int e = 0; e < 10; ++e) J = J + (n0 + e < N ? v[e] : T(0));
            }
            if (J != J) J = m_inf<T>();                  // NaN cost never wins the arg-min
            Jb[k0 + c] = J;
            Jd = (double)J; kk = k0 + c;
        }
        // wave arg-min, lowest index on ties (np.argmin): DPP row shifts when the candidates fit one
        // 16-lane row, ds_bpermute shuffles otherwise
        // branch-free: the three compares feed selects (a short-circuit form compiles to exec-mask branches per stage)
        auto take = [&](double oJ, long long ok) {
            const bool better = (oJ < Jd) | ((oJ == Jd) & (ok < kk));
            Jd = better ? oJ : Jd; kk = better ? ok : kk;
        };
        if (CK <= 16) {
            take(row_shl<8>(Jd), row_shl<8>(kk));
            take(row_shl<4>(Jd), row_shl<4>(kk));
            take(row_shl<2>(Jd), row_shl<2>(kk));
            take(row_shl<1>(Jd), row_shl<1>(kk));
        } else {
            for (int off = 32; off > 0; off >>= 1) take(__shfl_down(Jd, off, 64), __shfl_down(kk, off, 64));
        }
        if (c == 0) *s_best_c = (int)(kk - k0);
        if (fast_tail) {
            // the common case: this wave alone hands the workgroup's best over (lane 0 holds it after the reduction):
            // trajectory stores (write-through), drain, then the three tagged granules
            const int cb = __builtin_amdgcn_readfirstlane((int)(kk - k0));
            double *bt = blk_trajb + (size_t)blockIdx.x * (N + 1) * 2;
            if (c <= N) {
                st_agent(&bt[2 * c], (double)RV_PL(sY, 0, c, cb));
                st_agent(&bt[2 * c + 1], (double)RV_PL(sY, 1, c, cb));
            }
            if (SAMPLE) {
                double *bu = a.samp_blk_u + (size_t)blockIdx.x * 3 * N;
                for (int j = c; j < 3 * N; j += 64) st_agent(&bu[j], (double)sU[cb * US + j]);
            }
            if (c == 0) publish_best(granb, a.nblocks, a.epoch, Jd, (unsigned)cb);     // the cost does not wait for the drain
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (c == 0) publish_traj_ready(granb, a.nblocks, a.epoch);
        } else if (c == 0) {
            s_best_J = Jd;
        }
    }
    RV_STAMP(14);
    __syncthreads();
    RV_STAMP(15);
    if (!fast_tail) {
        const int cb = *s_best_c;
        double *bt = blk_trajb + (size_t)blockIdx.x * (N + 1) * 2;
        for (int i = tid; i < (N + 1); i += NT) {
            st_agent(&bt[2 * i], (double)RV_PL(sY, 0, i, cb));
            st_agent(&bt[2 * i + 1], (double)RV_PL(sY, 1, i, cb));
        }
        if (SAMPLE) {
            double *bu = a.samp_blk_u + (size_t)blockIdx.x * 3 * N;
            for (int j = tid; j < 3 * N; j += NT) st_agent(&bu[j], (double)sU[cb * US + j]);
        }
        if (trajb) {
            for (int i = tid; i < nvalid * (N + 1); i += NT) {
                const int c = i / (N + 1), n = i % (N + 1);
                T *dst = trajb + ((size_t)(k0 + c) * (N + 1) + n) * 2;
                dst[0] = RV_PL(sY, 0, n, c); dst[1] = RV_PL(sY, 1, n, c);
            }
        }
    }
    RV_STAMP(6);
    if (!resultb) return;
    if (!fast_tail) {

    // ---- arg-min epilogue in the sweeping workgroup ------------------------------------------------
    // Hand-off (cdna_hip_programming.md G16, form R2): every handed-off byte is stored write-through at
    // agent scope (sc1), the storing wave(s) drain, then ONE lane publishes the workgroup's best as
    // granules tagged with the launch epoch; one workgroup (a.sweeper) sweeps the granules with agent-scope loads
    // until all carry the epoch.  No assumption on dispatch order, timing or XCD placement.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) { publish_best(granb, a.nblocks, a.epoch, s_best_J, (unsigned)*s_best_c); publish_traj_ready(granb, a.nblocks, a.epoch); }
    }
    RV_STAMP(7);
    if ((int)blockIdx.x != a.sweeper) return;
    if (HANDOFF) {
        argmin_epilogue<T>(a, granb, blk_trajb, Ub, resultb, reinterpret_cast<double *>(smem + 4));
    } else if (LEAN) {
        argmin_epilogue<T, true>(a, granb, blk_trajb, Ub, resultb, reinterpret_cast<double *>(smem + 4));
    } else if (SAMPLE) {
        argmin_epilogue<T, false, true>(a, granb, blk_trajb, Ub, resultb, reinterpret_cast<double *>(smem + 4));
    } else {
        // The epilogue's own arguments (hand-off flags, slot buffer, plant update, host mirror ...) are cold: one workgroup
        // reads them once.  Read here through the kernarg segment pointer made opaque, their scalar loads cannot be
        // scheduled to the top of the kernel, where they would sit in SGPRs across the integration loop (measured: 27
        // more SGPR spill slots and +0.4 us per step with them loaded up front).  The argument struct is the kernel's
        // first (or only) parameter in every entry point, so it starts the segment.
#if defined(__HIP_DEVICE_COMPILE__)
        const RolloutArgs<T> *ap = (const RolloutArgs<T> *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ap));
#else
        const RolloutArgs<T> *ap = &a;
#endif
        argmin_epilogue<T>(*ap, granb, blk_trajb, Ub, resultb, reinterpret_cast<double *>(smem + 4));
    }
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel_long(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, false, false, true>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel_long_lean(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, true, false, true>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel_lean(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, true>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel16(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, false, false, false, 16>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel_long16(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, false, false, true, 16>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel_lean16(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, true, false, false, 16>(a);
}

// (and the horizon of the headline configuration, N = 20, as a literal as well: double precision only)
template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel_lean16_n20(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, true, false, false, 16, 20>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel16_n20(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, false, false, false, 16, 20>(a);
}

// (BASELINE configuration 3: N = 50, single precision)
template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel_long_lean16_n50(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, true, false, true, 16, 50>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel_long_lean16(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, true, false, true, 16>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
rollout_kernel_sampled(const RolloutArgs<T> a) {
    rollout_body<T, MODEL, VT, false, false, true>(a);
}

// ---- closed loop with the state handed over on the GPU ------------------------------------------------------------
// BASELINE config 5 on one GPU, pipelined form: ONE step per launch, launches alternating between two streams.  Launch
// g + 1 starts while launch g is still running -- its launch latency, dispatch ramp, control load, phase 2 and gamma chain
// leave the critical path -- and its workgroups wait on the step number for the state launch g's sweeper publishes
// (ring / seq_theta / seq_gamma, see RolloutArgs).  Both grids must be resident at once (the host checks it).  Every
// wait is bounded by a.handoff_ticks: a workgroup that gives up raises ERR_SWEEP and goes on, the sweeper's bounded
// sweep follows, and the grid drains.  (Round 2 also shipped the loop as ONE launch with the step loop inside the
// kernel; it was bit-equal and slower than this form -- 21 vs 13 us per step -- and was removed in round 3.)
struct HandoffArgs {
    long long T;                      // steps of the whole loop (the last step publishes no state)
    const double *exo;                // [T][16] measured rows (rovmpc_closed_loop_device)
    unsigned long long *seq_theta;    // steps whose (theta | gamma) record the sweepers have published (0 before the loop)
    unsigned long long *seq_gamma;
    double *ring;                     // [4][4], see RolloutArgs
    unsigned long long *granules2;    // [2][GRAN][nblocks] and
    double *blk_traj2;                // [2][nblocks][N+1][2]: hand-off buffers by step parity (two steps are in flight)
    long long step;                   // global index of this launch's step
};

// a.U / a.result / a.epoch are already this step's.
template <typename T, int MODEL, int VT, int CKC = 0, int NC = 0>
RV_DEV void closed_loop_step_body(const RolloutArgs<T> &a0, const HandoffArgs &p) {
    const long long g = p.step;
    RolloutArgs<T> a = a0;
    a.granules = p.granules2 + (size_t)(g & 1) * GRAN * a0.nblocks;
    a.blk_traj = p.blk_traj2 + (size_t)(g & 1) * a0.nblocks * (a0.N + 1) * 2;
    a.exo_cur = p.exo + (size_t)g * ROVMPC_STATE_LEN;
    a.ring = p.ring; a.seq_theta = p.seq_theta; a.seq_gamma = p.seq_gamma;
    a.step = g;
    a.from_ring = a0.plant_feedback && g > 0;
    a.wait_theta = a.from_ring;
    a.publish = g + 1 < p.T;
    a.plant_next = g + 1 < p.T ? p.exo + (size_t)(g + 1) * ROVMPC_STATE_LEN : nullptr;
    a.plant_state = const_cast<double *>(a0.state);
    rollout_body<T, MODEL, VT, true, false, false, false, CKC, NC>(a);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
closed_loop_step_kernel(const RolloutArgs<T> a, const HandoffArgs p) {
    closed_loop_step_body<T, MODEL, VT>(a, p);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
closed_loop_step_kernel16(const RolloutArgs<T> a, const HandoffArgs p) {
    closed_loop_step_body<T, MODEL, VT, 16>(a, p);
}

template <typename T, int MODEL, int VT>
__global__ void __launch_bounds__(512)
closed_loop_step_kernel16_n20(const RolloutArgs<T> a, const HandoffArgs p) {
    closed_loop_step_body<T, MODEL, VT, 16, 20>(a, p);
}

// After the all-reduce(min): every rank holds every rank's record; pick the lexicographic
// (cost, global index) minimum and decode it.
__global__ void __launch_bounds__(64)
select_kernel(const long long *slots, int world, int R, double *result, unsigned long long *flag_consumed = nullptr,
              unsigned long long consumed_seq = 0, const unsigned long long *slot_bad = nullptr, int inject = 0,
              double *ring = nullptr, unsigned long long *seq_theta = nullptr, long long step_next = 0) {
    __shared__ int s_r;
    if (threadIdx.x == 0) {
        double Jd = __builtin_inf(); double kd = __builtin_inf(); int rb = 0;
        for (int r = 0; r < world; ++r) {
            const double oJ = ordered_val(slots[(size_t)r * R]);
            const double ok = ordered_val(slots[(size_t)r * R + 1]);
            if (oJ < Jd || (oJ == Jd && ok < kd) || r == 0) { Jd = oJ; kd = ok; rb = r; }
        }
        s_r = rb;
    }
    __syncthreads();
    const int rb = s_r;
    // a hand-off of this use timed out (the row is not this step's): the record says so with a NaN cost
    const bool bad = slot_bad && ld_agent(slot_bad) == consumed_seq;
    for (int i = threadIdx.x; i < R; i += blockDim.x) result[i] = (bad && i == 0) ? __builtin_nan("") : ordered_val(slots[(size_t)rb * R + i]);
    // Sharded closed loop with the state handed over on the GPU: the rollout of step `step_next` is already launched and its
    // theta waves wait for this -- (theta, gamma) of its start = first predicted node of the GLOBAL winner, its delay slots =
    // what this step started from (record: [J, k, u(3), th0, ga0, th1, ga1, ...]); then the sequence word.
    if (ring && threadIdx.x == 0) {
        double *r4 = ring + (int)(step_next & 3) * 4;
        st_agent(&r4[0], ordered_val(slots[(size_t)rb * R + 7])); st_agent(&r4[1], ordered_val(slots[(size_t)rb * R + 8]));
        st_agent(&r4[2], ordered_val(slots[(size_t)rb * R + 5])); st_agent(&r4[3], ordered_val(slots[(size_t)rb * R + 6]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_agent(seq_theta, (unsigned long long)step_next);
    }
    if (flag_consumed && !(inject & 2)) {    // every read of the slot buffer is done: it may be rewritten
        __syncthreads();
        if (threadIdx.x == 0) st_agent(flag_consumed, consumed_seq);
    }
}

// Collective stream, ahead of the all-reduce of one step: wait until the rollout kernel of that step (running on the
// caller's stream) has published its row.  One lane polling with agent-scope loads; on giving up it raises the handle's
// error word and marks the use bad, so the step's record carries a NaN cost and rovmpc_comm_sync returns an error.
__global__ void __launch_bounds__(64)
wait_rolled_kernel(const unsigned long long *flag_rolled, unsigned long long seq, unsigned *err, unsigned long long *slot_bad,
                   unsigned long long handoff_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long give_up = wall_clock64() + handoff_ticks;
    for (;;) {
        if (ld_agent(flag_rolled) >= seq) return;
        if (wall_clock64() > give_up) break;
        __builtin_amdgcn_s_sleep(16);
    }
    raise_error(err, ERR_WAIT_ROLLED);
    st_agent(slot_bad, seq);
}

// State the closed loop with GPU-side hand-off leaves behind when the records, not the state buffer, carried it from step to
// step: the state step T - 1 started from (rows: [T][16]; records: [T][R]).
__global__ void __launch_bounds__(64)
final_state_kernel(double *state, const double *rows, const double *records, long long T, int R, int feedback) {
    const int t = threadIdx.x;
    if (t >= 16) return;
    const double *row = rows + (size_t)(T - 1) * 16;
    double v = row[t];
    if (feedback && T >= 2 && t >= 12) {
        const double *rec = records + (size_t)(T - 2) * R;
        v = t == 12 ? rec[7] : t == 13 ? rec[8] : t == 14 ? rec[5] : rec[6];
    }
    state[t] = v;
}

// Plant update of the closed-loop driver (one tiny workgroup): exogenous slots from the measured
// trajectory row, (theta, gamma) advanced to the first predicted node of the previous winner.
__global__ void __launch_bounds__(64)
plant_update_kernel(double *state, const double *exo_row, const double *prev_result) {
    const int t = threadIdx.x;
    if (prev_result) {          // feedback: keep the model's own (theta, gamma), take the rest from the row
        if (t < 12) state[t] = exo_row[t];
        if (t == 12) {
            const double th = state[12], ga = state[13];
            state[14] = th; state[15] = ga;
            state[12] = prev_result[7]; state[13] = prev_result[8];   // record: [J, k, u(3), th0, ga0, th1, ga1, ...]
        }
    } else if (t < 16) {
        state[t] = exo_row[t];
    }
}

}  // namespace rovmpc
