/* Plain-C consumer of librovmpc.so: proves the boundary is a C ABI (no Python, no torch).
 * Build: gcc -O2 -I include tests/c_abi/c_abi_smoke.c -o /tmp/c_abi_smoke -L <pkg>/lib -lrovmpc -lm
 * It checks solve_catenary against the reference's known answers, runs one MPC step with a
 * model given as bytecode, and checks the arg-min against the costs returned by the same library. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "rovmpc.h"

#define CHECK(h, call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rovmpc_last_error(h)); return 1; } } while (0)

int main(void) {
    rovmpc_config cfg;
    rovmpc_default_config(&cfg);
    cfg.N = 10; cfg.K = 128;
    rovmpc_handle *h = NULL;
    CHECK(NULL, rovmpc_create(&cfg, &h));

    /* main_fun.solve_catenary known answers (L = 3) */
    const double l[7] = {1.0, 1.41421356, 2.0, 2.5, 2.9, 0.5, 3.5}, dH[7] = {0, -1, 0.5, -0.3, 0.1, 0, 0};
    const double want[5] = {5.676892760096155, 3.0791940475045547, 1.5916068034624558, 0.8396630142778874, 0.309507897379277};
    double C[7], T[7];
    CHECK(h, rovmpc_solve_catenary(h, l, dH, 3.0, 7, C, T));
    for (int i = 0; i < 5; ++i)
        if (fabs(C[i] - want[i]) > 1e-11) { fprintf(stderr, "C[%d] = %.17g, want %.17g\n", i, C[i], want[i]); return 1; }
    if (!isnan(C[5]) || !isnan(C[6])) { fprintf(stderr, "expected NaN for the two no-root cases\n"); return 1; }
    if (fabs(T[5] - 1.521 / 3.0 * 0.5 / 2) > 1e-15) { fprintf(stderr, "tension fallback wrong\n"); return 1; }

    /* compute_catenary_3D (models/catenary_3d.py:5-39): a taut pair is the straight np.linspace, end points exact */
    {
        const double p0[3] = {0, 0, 0}, p1[3] = {3.0, 1.0, 0.5};
        double pts[4 * 3], a1;
        CHECK(h, rovmpc_compute_catenary_3d(h, p0, p1, 3.0, 1, 4, pts, &a1));
        if (!isnan(a1) || pts[9] != 3.0 || pts[10] != 1.0 || pts[11] != 0.5 || fabs(pts[3] - 1.0) > 1e-15) {
            fprintf(stderr, "compute_catenary_3d: straight case wrong\n"); return 1;
        }
    }

    /* model: dtheta/dt = -0.05 * x16 - 0.05 * sin(x3), dgamma/dt = x15 - x17 (identity scaler) */
    double mean[18] = {0}, scale[18];
    for (int i = 0; i < 18; ++i) scale[i] = 1.0;
    const double consts[1] = {-0.05};
    const int32_t th[] = {(0 << 8) | ROVMPC_OP_PUSH_C, (16 << 8) | ROVMPC_OP_PUSH_F, ROVMPC_OP_MUL,
                          (0 << 8) | ROVMPC_OP_PUSH_C, (3 << 8) | ROVMPC_OP_PUSH_F, ROVMPC_OP_SIN, ROVMPC_OP_MUL, ROVMPC_OP_ADD};
    const int32_t ga[] = {(15 << 8) | ROVMPC_OP_PUSH_F, (17 << 8) | ROVMPC_OP_PUSH_F, ROVMPC_OP_SUB};
    CHECK(h, rovmpc_set_model(h, 18, mean, scale, th, 8, ga, 3, consts, 1));
    printf("model path: %d (0 compiled-in, 1 interpreter, 2 hiprtc)\n", rovmpc_model_path(h));

    rovmpc_state st = {{0, 0, 0}, {0.24, -0.76, 0.30}, {0.5, -0.2, 0.1}, {0, 0, 0}, -0.03, -0.05, -0.03, -0.05};
    double *U = malloc(sizeof(double) * cfg.K * cfg.N * 3), *J = malloc(sizeof(double) * cfg.K);
    unsigned s = 12345u;
    for (int i = 0; i < cfg.K * cfg.N * 3; ++i) { s = s * 1664525u + 1013904223u; U[i] = ((double)(s >> 8) / 16777216.0 - 0.5) * 4.0; }
    double u[3], traj[2 * 11], best; int64_t idx;
    CHECK(h, rovmpc_step(h, &st, U, u, traj, &best, &idx));
    CHECK(h, rovmpc_rollout_costs(h, &st, U, J, NULL));
    int k = 0;
    for (int i = 1; i < cfg.K; ++i) if (J[i] < J[k]) k = i;
    if (k != idx || J[k] != best || u[0] != U[(size_t)k * cfg.N * 3]) { fprintf(stderr, "arg-min mismatch: %d vs %lld\n", k, (long long)idx); return 1; }
    if (traj[0] != st.theta || traj[1] != st.gamma) { fprintf(stderr, "trajectory does not start at the state\n"); return 1; }

    /* round-2 entry points, still plain C: one control step with the candidates drawn on the GPU (same (seed, step) ->
     * same record; the tensor can be fetched and its arg-min must be the record's), a program evaluated on rows
     * (the Euler-Lagrange residual 2 x4 of L = x2^2 + x3^2), the handle's error word */
    const double mean3[3] = {0.0, 0.0, 0.0}, std3[3] = {2.0, 2.0, 2.0};
    const int R = rovmpc_result_len(h);
    double *rec = malloc(sizeof(double) * R), *rec2 = malloc(sizeof(double) * R);
    CHECK(h, rovmpc_mpc_step_sampled(h, &st, 7u, 0u, mean3, std3, 0, rec));
    CHECK(h, rovmpc_sampled_candidates(h, U));
    CHECK(h, rovmpc_rollout_costs(h, &st, U, J, NULL));
    k = 0;
    for (int i = 1; i < cfg.K; ++i) if (J[i] < J[k]) k = i;
    if ((int)rec[1] != k || rec[0] != J[k] || rec[2] != U[(size_t)k * cfg.N * 3]) { fprintf(stderr, "sampled step: arg-min mismatch\n"); return 1; }
    rovmpc_handle *h2 = NULL;
    CHECK(NULL, rovmpc_create(&cfg, &h2));
    CHECK(h2, rovmpc_set_model(h2, 18, mean, scale, th, 8, ga, 3, consts, 1));
    CHECK(h2, rovmpc_mpc_step_sampled(h2, &st, 7u, 0u, mean3, std3, 0, rec2));
    for (int i = 0; i < R; ++i) if (rec[i] != rec2[i]) { fprintf(stderr, "sampled step is not a function of (state, seed, step)\n"); return 1; }
    rovmpc_destroy(h2);
    const double two[1] = {2.0};
    const int32_t eom[] = {(0 << 8) | ROVMPC_OP_PUSH_C, (4 << 8) | ROVMPC_OP_PUSH_F, ROVMPC_OP_MUL};
    const double rows[2 * 6] = {0.1, 0.2, 0.3, 0.4, -1.5, 0.6, 0, 0, 0, 0, 0.25, 0};
    double res[2];
    CHECK(h, rovmpc_eval_expression(h, eom, 3, two, 1, rows, 6, 2, res));
    if (res[0] != -3.0 || res[1] != 0.5) { fprintf(stderr, "eval_expression: %g %g\n", res[0], res[1]); return 1; }
    CHECK(h, rovmpc_device_status(h));
    if (rovmpc_set_option(h, "no_such_option", 1.0) != ROVMPC_ERR_INVALID) { fprintf(stderr, "unknown option accepted\n"); return 1; }
    free(rec); free(rec2);
    printf("c_abi_smoke ok: k*=%lld J*=%.12g u=(%.6f %.6f %.6f)\n", (long long)idx, best, u[0], u[1], u[2]);
    free(U); free(J);
    rovmpc_destroy(h);
    return 0;
}
