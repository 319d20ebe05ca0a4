"""CPU: the oracle (oracle/rovmpc_oracle.py) against the golden vectors generated from the
reference itself (tools/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest

from oracle import rovmpc_oracle as orc


def test_scaler_anchor_values(scaler):
    mean, scale = scaler
    assert mean.shape == (18,) and scale.shape == (18,)
    assert mean[3] == 80.85390753943355 and scale[17] == 0.017328840568644522


def test_equations_chosen_rows(equations):
    assert equations["dtheta_dt"]["chosen_complexity"] == 13
    assert equations["dgamma_dt"]["chosen_complexity"] == 3
    assert len(equations["dtheta_dt"]["rows"]) == 23 and len(equations["dgamma_dt"]["rows"]) == 14


def test_dynamics_all_rows(golden_dir, equations):
    g = np.load(os.path.join(golden_dir, "kat_dynamics.npz"))
    Xs = g["Xs"]
    for which, key in (("dtheta_dt", "out_theta"), ("dgamma_dt", "out_gamma")):
        for i, row in enumerate(equations[which]["rows"]):
            m = orc.SymbolicModel(row["sympy_format"])
            np.testing.assert_allclose(m.predict(Xs), g[key][i], rtol=1e-13, atol=1e-16)


def test_survey_row0_kat(equations):
    xs = np.zeros((1, 18))
    xs[0, 3], xs[0, 15], xs[0, 16], xs[0, 17] = (0.10490011715303967, -0.7322673547034521,
                                                 -0.5442589828573099, -0.31630015636915465)
    from conftest import chosen_row
    th = orc.SymbolicModel(chosen_row(equations, "dtheta_dt")["sympy_format"]).predict(xs)[0]
    ga = orc.SymbolicModel(chosen_row(equations, "dgamma_dt")["sympy_format"]).predict(xs)[0]
    assert th == pytest.approx(0.0011363337060931117, rel=1e-12)
    assert ga == pytest.approx(-0.4159671983342974, rel=1e-12)


def test_solve_catenary_brentq_and_tension(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_solve_catenary.npz"))
    C = orc.solve_catenary_ref(g["l"], g["dH"], float(g["L"]))
    assert np.array_equal(np.isnan(C), np.isnan(g["C"]))
    np.testing.assert_allclose(C, g["C"], rtol=0, atol=0, equal_nan=True)       # same brentq
    T = orc.cable_tension(g["l"], C, float(g["L"]), float(g["w_wet"]))
    np.testing.assert_allclose(T, g["T"], rtol=1e-14)
    # SURVEY KATs
    kat = {(1.0, 0): 5.676892760096155, (1.41421356, -1): 3.0791940475045547,
           (2.0, 0.5): 1.5916068034624558, (2.5, -0.3): 0.8396630142778874,
           (2.9, 0.1): 0.309507897379277}
    for (l, dh), c in kat.items():
        assert orc.solve_catenary_scalar(l, dh, 3.0) == pytest.approx(c, rel=1e-12)
    assert np.isnan(orc.solve_catenary_scalar(0.5, 0, 3.0))
    assert np.isnan(orc.solve_catenary_scalar(3.5, 0, 3.0))


def test_solve_catenary_vec_matches_brentq(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_solve_catenary.npz"))
    rng = np.random.default_rng(5)
    l = np.concatenate([g["l_in"], rng.uniform(1e-3, 3.5, 2000)])
    dH = np.concatenate([g["dH_in"], rng.uniform(-3.2, 3.2, 2000)])
    ref = orc.solve_catenary_ref(l, dH, 3.0)
    vec = orc.solve_catenary_vec(l, dH, 3.0)
    assert np.array_equal(np.isnan(ref), np.isnan(vec))
    np.testing.assert_allclose(vec, ref, rtol=0, atol=1e-11, equal_nan=True)
    assert np.isfinite(ref).sum() > 300 and np.isnan(ref).sum() > 300


def test_rodrigues(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_rodrigues.npz"))
    for i in range(len(g["angle"])):
        np.testing.assert_allclose(orc.rodrigues_rotation(g["v"][i], g["axis"][i], g["angle"][i]),
                                   g["out"][i], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(g["out"][0], [0.86201995, 0.86201995, -1.23038336], atol=5e-9)


def test_transform_catenary(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_transform_catenary.npz"))
    cat = orc.Catenary(float(g["L"]), "ENU", n_points=int(g["M"]))
    n_none = 0
    for i in range(len(g["theta"])):
        r = orc.transform_catenary(g["A"][i], g["B"][i], cat, g["theta"][i], g["gamma"][i])
        n0, n1 = g["npts"][i]
        n_none += int(n1 == 2)
        for out, key, n in zip(r, ("original", "theta_rotated", "theta_aligned", "final"),
                               (n0, n1, n1, n1)):
            assert out.shape == (n, 3)
            np.testing.assert_allclose(out, g[key][i][:n], rtol=1e-13, atol=1e-14)
    assert n_none >= 2      # the straight-segment fallback was exercised


def test_compute_catenary_3d_is_the_reference_function(golden_dir):
    """The in-repo catenary generator (models/catenary_3d.py:5-39): the restatement against the outputs of the reference's
    own function (tools/make_golden_catenary3d.py), hanging, taut and nearly taut pairs."""
    g = np.load(os.path.join(golden_dir, "kat_catenary_3d.npz"))
    M = int(g["M"])
    straight = 0
    for i in range(len(g["rope"])):
        pts = orc.compute_catenary_3d(g["p0"][i], g["p1"][i], float(g["rope"][i]), M)
        np.testing.assert_allclose(pts, g["points"][i], rtol=1e-13, atol=1e-14)
        straight += int(np.linalg.norm(g["p1"][i] - g["p0"][i]) >= g["rope"][i])
    assert straight >= 2                                    # the np.linspace branch was exercised
    # the script's own demo pair, 12 m of rope over 10 m.  The reference's update a <- a L / (2 a sinh(d / 2a)) multiplies a
    # by L / arc > 1 for as long as the rope is longer than the arc a implies, so it runs its 100 rounds and ends on a
    # nearly flat curve (sag 1e-7 m here): that behaviour IS the function, and it is reproduced as it is
    mid = g["points"][0][M // 2]
    assert 0.0 < 10.0 - mid[2] < 1e-6


def test_transform_catenary_on_reference_code_alone(golden_dir):
    """main_fun.transform_catenary with the reference's compute_catenary_3D as catenary_fn: no stand-in for the absent
    pympc class anywhere in the path."""
    g = np.load(os.path.join(golden_dir, "kat_transform_catenary_3d.npz"))
    cat = orc.Catenary3D(float(g["L"]), int(g["M"]))
    for i in range(len(g["theta"])):
        r = orc.transform_catenary(g["A"][i], g["B"][i], cat, g["theta"][i], g["gamma"][i])
        for out, key in zip(r, ("original", "theta_rotated", "theta_aligned", "final")):
            np.testing.assert_allclose(out, g[key][i], rtol=1e-12, atol=1e-13)


def test_lowest_z_vec_matches_transform_catenary(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_transform_catenary.npz"))
    M = int(g["M"])
    for i in range(len(g["theta"])):
        n1 = g["npts"][i][1]
        zref = np.min(g["final"][i][:n1, 2])
        z = orc.augmented_lowest_z_vec(g["A"][i], g["B"][i][None, :], g["theta"][i:i + 1],
                                       g["gamma"][i:i + 1], 3.0, M, 1.0, 1e-6, 10.0)
        assert z[0] == pytest.approx(zref, rel=1e-12, abs=1e-13)


def test_catenary_arc_length_and_endpoints():
    cat = orc.Catenary(3.0, "ENU", n_points=2001)
    a = np.array([0.1, -0.2, 0.3]); b = np.array([1.2, 0.7, -0.4])
    C, sag, x0, pts = cat(a, b)
    seg = np.linalg.norm(np.diff(pts, axis=0), axis=1).sum()
    assert seg == pytest.approx(3.0, rel=1e-6)
    np.testing.assert_allclose(pts[0], a, atol=1e-14)
    np.testing.assert_allclose(pts[-1], b, atol=1e-12)
    assert pts[:, 2].min() < min(a[2], b[2])
    ned = orc.Catenary(3.0, "NED", n_points=64)(a, b)[3]
    assert ned[:, 2].max() > max(a[2], b[2])
    assert orc.Catenary(3.0)(a, a + np.array([3.0, 0, 0]))[3] is None


def test_feature_maps(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_features.npz"))
    fr = g["frame"]
    P0 = fr[:, 0:3] / 1000; P1 = fr[:, 3:6] / 1000; V1 = fr[:, 6:9]
    th, ga, t = fr[:, 9], fr[:, 10], fr[:, 11]
    X = orc.extract_features_gen1(P0, P1, V1, t, th, ga)
    np.testing.assert_allclose(X, g["X18"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(X[:, :16], g["X16"], rtol=1e-13, atol=1e-15)


def test_replay_integrators(golden_dir, equations):
    from conftest import chosen_row
    g = np.load(os.path.join(golden_dir, "kat_replay.npz"))
    mt = orc.SymbolicModel(chosen_row(equations, "dtheta_dt")["sympy_format"])
    mg = orc.SymbolicModel(chosen_row(equations, "dgamma_dt")["sympy_format"])
    mt2 = orc.SymbolicModel(equations["dtheta_dt"]["rows"][-1]["sympy_format"])
    mg2 = orc.SymbolicModel(equations["dgamma_dt"]["rows"][-1]["sympy_format"])
    Xs, t = g["Xs"], g["time"]
    th0, ga0 = float(g["theta0"]), float(g["gamma0"])
    np.testing.assert_allclose(orc.rk4_replay(mt.predict, Xs, t, th0), g["rk4_theta"], rtol=1e-12)
    np.testing.assert_allclose(orc.rk4_replay(mg.predict, Xs, t, ga0), g["rk4_gamma"], rtol=1e-12)
    np.testing.assert_allclose(orc.rk4_replay(mt2.predict, Xs, t, th0), g["rk4_theta_last"], rtol=1e-12)
    np.testing.assert_allclose(orc.rk4_replay(mg2.predict, Xs, t, ga0), g["rk4_gamma_last"], rtol=1e-12)
    e = orc.euler_replay(mt.predict, mg.predict, Xs, t, th0, ga0)
    np.testing.assert_allclose(np.stack(e), g["euler"], rtol=1e-12)
    e2 = orc.euler_replay(mt2.predict, mg2.predict, Xs, t, th0, ga0)
    np.testing.assert_allclose(np.stack(e2), g["euler_last"], rtol=1e-12)


@pytest.mark.parametrize("case", [1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 13, 14])
def test_rov_trajectory_generator(golden_dir, case):
    lines = open(os.path.join(golden_dir, f"rov_trajectory_exp{case}.csv")).read().strip().split("\n")
    _, t0, t1 = orc.rov_trajectories(case, n_steps=100, total_time=10.0)
    rows = orc.rov_trajectory_csv_rows(t0, t1)
    assert len(lines) == 101
    got = np.array([[float(v) for v in r.split(",")] for r in rows])
    want = np.array([[float(v) for v in r.split(",")] for r in lines[1:]])
    # the reference rounds with %.3f; a half-ulp tie may print either way across libm versions
    np.testing.assert_allclose(got, want, atol=1.0001e-3, rtol=0)
    assert (got != want).mean() < 0.01


def test_closed_loop_reduces_to_reference_rk4_when_f_ignores_state(scaler):
    """A3: with an f that reads only exogenous slots the closed-loop rollout must equal
    simulate_rk4_theta_gamma.py:52-68 applied to the same feature rows."""
    mean, scale = scaler
    f_th = orc.SymbolicModel("0.3*sin(x3) - 0.1*x4*x13 + 0.05*x0")
    f_ga = orc.SymbolicModel("0.2*x5 - 0.1*cos(x9) + x12*0.01")
    model = orc.DynamicsModel(mean, scale, f_th, f_ga)
    cfg = orc.MPCConfig(N=12, dt=0.02, vt_mode=0)
    rng = np.random.default_rng(1)
    st = orc.MPCState(np.zeros(3), np.array([0.24, -0.76, 0.3]), np.array([80., -20., -18.]),
                      np.zeros(3), -0.03, -0.05, -0.03, -0.05)
    U = mean[3:6] + scale[3:6] * rng.standard_normal((3, cfg.N, 3))
    J, traj, _ = orc.rollout_scalar(cfg, model, st, U)
    for k in range(3):
        # rebuild the exogenous feature table the rollout saw, then run the reference formula
        P = [st.P1]; V = [st.V1]; A = [st.A1]
        for n in range(cfg.N):
            P.append(P[-1] + cfg.v_scale * cfg.dt * U[k, n]); V.append(U[k, n])
            A.append((V[-1] - V[-2]) / cfg.dt)
        rows = np.stack([np.concatenate([orc._exo_features_scalar(st.P0, P[i], V[i], A[i], mean, scale),
                                         np.zeros(4)]) for i in range(cfg.N + 1)])
        t = np.arange(cfg.N + 1) * cfg.dt
        np.testing.assert_allclose(traj[k, :, 0], orc.rk4_replay(f_th.predict, rows, t, st.theta),
                                   rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(traj[k, :, 1], orc.rk4_replay(f_ga.predict, rows, t, st.gamma),
                                   rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("vt_mode,prev_mode,integrator", [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 0, 1), (2, 0, 0)])
def test_scalar_and_vector_flavours_agree(oracle_model, vt_mode, prev_mode, integrator):
    cfg = orc.MPCConfig(N=8, vt_mode=vt_mode, prev_mode=prev_mode, integrator=integrator,
                        n_shape_pts=8)
    rng = np.random.default_rng(2)
    mean, scale = oracle_model.mean, oracle_model.scale
    st = orc.MPCState(np.zeros(3), mean[0:3] + 0.05 * rng.standard_normal(3), mean[3:6].copy(),
                      np.zeros(3), mean[14], mean[15], mean[14], mean[15])
    U = mean[3:6] + scale[3:6] * rng.standard_normal((12, cfg.N, 3))
    Rtab = None
    if vt_mode == 2:
        Rtab = np.stack([np.linalg.qr(rng.standard_normal((3, 3)))[0] for _ in range(cfg.N)])
    Js, ts, auxs = orc.rollout_scalar(cfg, oracle_model, st, U, Rtab)
    Jv, tv, auxv = orc.rollout_vec(cfg, oracle_model, st, U, Rtab)
    np.testing.assert_allclose(tv, ts, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(auxv["C"], auxs["C"], rtol=1e-10, equal_nan=True)
    np.testing.assert_allclose(auxv["T"], auxs["T"], rtol=1e-10)
    np.testing.assert_allclose(auxv["z_low"], auxs["z_low"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(Jv, Js, rtol=1e-10)
    assert int(np.argmin(Jv)) == int(np.argmin(Js))


def test_second_order_replays_are_consistent():
    """Constant second derivative a: double Euler gives y0 + a dt^2 i(i-1)/2, trapezoid gives
    y0 + a dt^2 i(i+1)/2 (the reference's cumsum uses the END-of-step velocity)."""
    t = np.arange(11) * 0.1
    a = np.full(11, 2.0)
    th, ga = orc.double_euler_replay(a, -a, t, 1.0, -1.0)
    i = np.arange(11)
    np.testing.assert_allclose(th, 1.0 + 2.0 * 0.01 * i * (i - 1) / 2, rtol=1e-12)
    np.testing.assert_allclose(ga, -th, rtol=1e-12)
    th, _ = orc.trapezoid_replay(a, a, t, 1.0, 1.0)
    np.testing.assert_allclose(th, 1.0 + 2.0 * 0.01 * i * (i + 1) / 2, rtol=1e-12)


def test_kabsch_rotation(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_kabsch.npz"))
    for i in range(len(g["v"])):
        R = orc.compute_rotation_kabsch(g["P"][i].copy(), g["Q"][i].copy())
        np.testing.assert_allclose(R, g["R"][i], rtol=0, atol=1e-12)
        np.testing.assert_allclose(R.T @ R, np.eye(3), atol=1e-12)
        assert np.linalg.det(R) == pytest.approx(1.0, abs=1e-12)
    v, R = orc.kabsch_velocity_transform(g["P"], g["Q"], g["v"])
    np.testing.assert_allclose(v, g["v_out"], rtol=1e-11, atol=1e-11)


def test_generation2_equations(golden_dir):
    import json
    eq = json.load(open(os.path.join(golden_dir, "equations_gen2.json")))
    assert eq["dtheta_dt"]["chosen_complexity"] == 13 and eq["dgamma_dt"]["chosen_complexity"] == 20
    g = np.load(os.path.join(golden_dir, "kat_dynamics_gen2.npz"))
    for which, key in (("dtheta_dt", "out_theta"), ("dgamma_dt", "out_gamma")):
        for i, row in enumerate(eq[which]["rows"]):
            got = orc.SymbolicModel(row["sympy_format"], 17).predict(g["X"])
            np.testing.assert_allclose(got, g[key][i], rtol=1e-12, atol=1e-14, equal_nan=True)


def test_generation2_scalar_and_vector_rollouts_agree(golden_dir):
    import json
    eq = json.load(open(os.path.join(golden_dir, "equations_gen2.json")))
    row = lambda w: [r for r in eq[w]["rows"] if r["complexity"] == eq[w]["chosen_complexity"]][0]["sympy_format"]
    model = orc.DynamicsModel(np.zeros(17), np.ones(17), orc.SymbolicModel(row("dtheta_dt"), 17), orc.SymbolicModel(row("dgamma_dt"), 17))
    cfg = orc.MPCConfig(N=6, n_shape_pts=6, feature_map=1)
    rng = np.random.default_rng(3)
    st = orc.MPCState(np.zeros(3), np.array([0.24, -0.76, 0.3]), np.array([80., -20., -18.]), np.zeros(3), -0.03, -0.05, -0.03, -0.05)
    U = np.array([80., -20., -18.]) + np.array([100., 15., 60.]) * rng.standard_normal((10, 6, 3))
    Js, ts, _ = orc.rollout_scalar(cfg, model, st, U)
    Jv, tv, _ = orc.rollout_vec(cfg, model, st, U)
    np.testing.assert_allclose(tv, ts, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(Jv, Js, rtol=1e-10)


# ---- second-order generation (features_dd, dd_cluster.py) -----------------------------------

def _gen3(golden_dir):
    import json
    eq = json.load(open(os.path.join(golden_dir, "equations_gen3.json")))
    sc = json.load(open(os.path.join(golden_dir, "scaler_gen3.json")))
    return eq, np.array(sc["mean"]), np.array(sc["scale"])


def test_features_dd_equal_the_reference(golden_dir):
    """oracle.features_dd vs main_fun.features_dd run on the same synthetic log (savgol 11/3, gradient chains)."""
    d = np.load(os.path.join(golden_dir, "kat_features_dd.npz"))
    F, Y = orc.features_dd(d["P0"], d["P1"], d["V"], d["time"], d["theta"], d["gamma"])
    np.testing.assert_allclose(F, d["features"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(Y, d["targets"], rtol=1e-12, atol=1e-14)


def test_generation3_rows_named_variables(golden_dir):
    eq, mean, scale = _gen3(golden_dir)
    assert eq["variable_names"][1] == "gama" and len(mean) == 14           # dd_cluster.py:160-168
    assert eq["ddtheta"]["chosen_complexity"] == 6 and eq["ddgamma"]["chosen_complexity"] == 5
    g = np.load(os.path.join(golden_dir, "kat_dynamics_gen3.npz"))
    for which, key in (("ddtheta", "out_theta"), ("ddgamma", "out_gamma")):
        for i, row in enumerate(eq[which]["rows"]):
            m = orc.SymbolicModel(row["sympy_format"], 14, eq["variable_names"])
            np.testing.assert_allclose(m.predict(g["X"]), g[key][i], rtol=1e-13, atol=1e-16)


@pytest.mark.parametrize("vt_mode,integrator", [(0, 0), (1, 0), (1, 1), (2, 0)])
def test_second_order_rollout_scalar_equals_vectorised(golden_dir, vt_mode, integrator):
    eq, mean, scale = _gen3(golden_dir)
    pick = lambda w: [r for r in eq[w]["rows"] if r["complexity"] == eq[w]["chosen_complexity"]][0]["sympy_format"]  # noqa: E731
    model = orc.DynamicsModel(mean, scale, orc.SymbolicModel(pick("ddtheta"), 14, eq["variable_names"]),
                              orc.SymbolicModel(pick("ddgamma"), 14, eq["variable_names"]))
    rng = np.random.default_rng(5)
    K, N = 5, 9
    U = np.array([80.85, -20.13, -18.35]) + np.array([108.49, 15.88, 63.13]) * rng.standard_normal((K, N, 3))
    st = orc.MPCState(np.zeros(3), np.array([0.2435, -0.7583, 0.2980]), np.array([80.85, -20.13, -18.35]),
                      np.array([5., -3., 2.]), -0.0342, -0.0522, 0.01, -0.02)
    Rtab = None
    if vt_mode == 2:
        Rtab = np.stack([np.linalg.qr(rng.standard_normal((3, 3)))[0] for _ in range(N)])
    cfg = orc.MPCConfig(N=N, dt=1 / 60, vt_mode=vt_mode, integrator=integrator, feature_map=2)
    J, traj, aux = orc.rollout_vec_dd(cfg, model, st, U, Rtab)
    ts = orc.rollout_scalar_dd(cfg, model, st, U, Rtab)
    np.testing.assert_allclose(traj, ts, rtol=1e-12, atol=1e-15)
    assert np.all(np.isfinite(J))
    # double Euler of the reference on a frozen input equals its open-loop replay (test_cluster.py:110-129)
    if integrator == 1 and vt_mode == 0:
        zero = orc.DynamicsModel(mean, scale, orc.SymbolicModel("0.25 + 0*theta", 14, eq["variable_names"]),
                                 orc.SymbolicModel("-0.5 + 0*theta", 14, eq["variable_names"]))
        _, tz, _ = orc.rollout_vec_dd(cfg, zero, st, U[:1])
        t = np.arange(N + 1) * cfg.dt
        th, ga = orc.double_euler_replay(np.full(N + 1, 0.25), np.full(N + 1, -0.5), t, st.theta, st.gamma)
        # the replay starts from zero rates (test_cluster.py:111-112); add the initial-rate ramp
        np.testing.assert_allclose(tz[0, :, 0], th + st.theta_prev * t, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(tz[0, :, 1], ga + st.gamma_prev * t, rtol=1e-12, atol=1e-15)


def test_philox_restatement_known_answers():
    """oracle.philox4x32_10 against the published Random123 known-answer vectors (kat_vectors: philox4x32 10), and the
    moments of the normals it feeds (the proposal law of MPC.step with device sampling)."""
    from oracle import rovmpc_oracle as orc
    u32 = lambda *v: [np.array([x], np.uint32) for x in v]
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = orc.philox4x32_10(*u32(*ctr), *key)
        assert tuple(int(g[0]) for g in got) == want
    z = orc.philox_normals(7, 3, 400001)
    assert z.shape == (400001,) and abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert np.array_equal(orc.philox_normals(7, 3, 10), z[:10]) and not np.array_equal(orc.philox_normals(7, 4, 10), z[:10])
    U = orc.sample_candidates(7, 3, 5, 4, [1.0, 2.0, 3.0], [0.5, 0.25, 2.0], prev_best=np.arange(12.0).reshape(4, 3))
    assert np.array_equal(U[0], np.array([[3, 4, 5], [6, 7, 8], [9, 10, 11], [9, 10, 11]], float))
    assert np.allclose(U[1, 0], np.array([1.0, 2.0, 3.0]) + np.array([0.5, 0.25, 2.0]) * z[12:15])


def test_lagrangian_oracle_reproduces_the_reference_residual_files(golden_dir):
    """oracle.el_residuals (sympy, the reference's route) against the residual series the reference's own Lagrangian runs
    stored (outputs/Lg_C6_*/euler_lagrange_residuals.npz on trajectory_data.npz), incl. the two split runs whose T - V
    cancels identically."""
    from oracle import rovmpc_oracle as orc
    g = np.load(os.path.join(golden_dir, "kat_lagrangian.npz"))
    series = [g[k] for k in ("theta", "gamma", "dtheta", "dgamma", "ddtheta", "ddgamma")]
    for tag in ("full", "split_hy", "split"):
        r_th, r_ga = orc.el_residuals(str(g[f"expr_{tag}"]), *series)
        np.testing.assert_allclose(r_th, g[f"residual_theta_{tag}"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(r_ga, g[f"residual_gamma_{tag}"], rtol=1e-12, atol=1e-15)
    assert np.allclose(g["residual_theta_full"], 2 * g["ddtheta"])               # L = dtheta^2 + dgamma^2
