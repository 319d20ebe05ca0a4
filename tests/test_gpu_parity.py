"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden vectors.

Tolerances: the bar set by BASELINE.json is 1e-5 relative (fp64) on the chosen u and the
predicted (theta, gamma) trajectory; these tests hold the kernels to 1e-9 (fp64) because
nothing in the path amplifies rounding at the benchmark's step size, and to the fp32 rule of
SURVEY section 8(d) (same k* or |J32 - J64| / J64 < 1e-4) for the fp32 variant.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.fixture(scope="module")
def rv():
    import rovmpc
    return rovmpc


@pytest.fixture(scope="module")
def orc():
    from oracle import rovmpc_oracle
    return rovmpc_oracle


def oracle_cfg(orc, cfg):
    return orc.MPCConfig(N=cfg.N, dt=cfg.dt, v_scale=cfg.v_scale, L=cfg.L, cable_wet_weight=cfg.cable_wet_weight,
                         c_lo=cfg.c_lo, c_hi=cfg.c_hi, n_shape_pts=cfg.n_shape_pts,
                         up=1.0 if cfg.frame == "ENU" else -1.0, vt_mode=cfg.vt_mode, prev_mode=cfg.prev_mode,
                         integrator=cfg.integrator, w_theta=cfg.w_theta, w_gamma=cfg.w_gamma, w_u=cfg.w_u,
                         w_T=cfg.w_T, w_taut=cfg.w_taut, rho_taut=cfg.rho_taut, w_floor=cfg.w_floor,
                         z_floor=cfg.z_floor, theta_ref=cfg.theta_ref, gamma_ref=cfg.gamma_ref, U_ref=tuple(cfg.U_ref),
                         feature_map=cfg.feature_map)


def oracle_model(orc, model):
    return orc.DynamicsModel(model.mean, model.scale, orc.SymbolicModel(model.expr_theta),
                             orc.SymbolicModel(model.expr_gamma))


def rand_rtab(N, seed=7):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(N):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        out.append(q)
    return np.stack(out)


# ---------------------------------------------------------------------------------------------
# helper mirrors vs golden vectors
# ---------------------------------------------------------------------------------------------

def test_library_is_the_hip_one(rv):
    import ctypes
    lib = rv.load_library()
    assert b"gfx950" in lib.rovmpc_version()
    assert os.path.samefile(lib._name, rv.LIB_PATH)


def test_predict_every_pareto_row(rv, golden_dir, equations, scaler):
    g = np.load(os.path.join(golden_dir, "kat_dynamics.npz"))
    mean, scale = scaler
    rows_t = equations["dtheta_dt"]["rows"]; rows_g = equations["dgamma_dt"]["rows"]
    for i in range(max(len(rows_t), len(rows_g))):
        rt = rows_t[min(i, len(rows_t) - 1)]; rg = rows_g[min(i, len(rows_g) - 1)]
        m = rv.DynamicsModel(mean, scale, rt["sympy_format"], rg["sympy_format"])
        with rv.Engine(rv.MPCConfig(N=1, K=1, force_interpreter=True), m) as e:
            np.testing.assert_allclose(e.predict(g["Xs"], 0), g["out_theta"][min(i, len(rows_t) - 1)], rtol=1e-12, atol=1e-15)
            np.testing.assert_allclose(e.predict(g["Xs"], 1), g["out_gamma"][min(i, len(rows_g) - 1)], rtol=1e-12, atol=1e-15)


def test_hoisted_subexpressions_do_not_change_the_rollout(rv):
    """The hiprtc path evaluates a model's stage-invariant subexpressions once per row instead of once per RK4 stage
    (bytecode_to_cxx / jit_exo).  With the hoisting switched off (ROVMPC_JIT_NO_HOIST) the same models give the same costs and
    the same winner, to rounding: a hoisted product no longer contracts into the FMA around it, a hoisted divisor is applied as
    a reciprocal."""
    cases = [("gen2", rv.generation2_model(), rv.FEATURES_GEN2), ("gen3", rv.generation3_model(), rv.FEATURES_GEN3),
             ("reference rows", rv.default_model(), rv.FEATURES_GEN1), ("rows 16/20", rv.default_model(16, 20), rv.FEATURES_GEN1)]
    for name, model, fmap in cases:
        out = {}
        for hoist in (True, False):
            if hoist:
                os.environ.pop("ROVMPC_JIT_NO_HOIST", None)
            else:
                os.environ["ROVMPC_JIT_NO_HOIST"] = "1"
            try:
                cfg = rv.MPCConfig(N=20, K=512, feature_map=fmap, no_builtin=True)
                state, U = rv.synthetic_problem(cfg.K, cfg.N, seed=77)
                with rv.Engine(cfg, model) as e:
                    assert e.model_path == "jit", name
                    out[hoist] = (e.step(state, U), e.rollout_costs(state, U))
            finally:
                os.environ.pop("ROVMPC_JIT_NO_HOIST", None)
        (ra, Ja), (rb, Jb) = out[True], out[False]
        assert ra.index == rb.index, name
        np.testing.assert_allclose(Ja, Jb, rtol=1e-11, err_msg=name)
        np.testing.assert_allclose(ra.traj, rb.traj, rtol=1e-11, atol=1e-14, err_msg=name)


def test_expression_division_keeps_numpy_special_cases(rv):
    """A loaded model's `/` is a reciprocal with two Newton steps, a residual correction and v_div_fixup (m_divq) instead of
    the eleven-instruction IEEE sequence: regular quotients to 1 ulp, and x / 0 = +-inf, 0 / 0 = NaN, x / inf = 0, NaN in ->
    NaN out, exactly as NumPy evaluates the reference's lambdified rows."""
    rng = np.random.default_rng(3)
    X = np.zeros((4096, 18))
    X[:, 0] = rng.standard_normal(4096) * 10.0 ** rng.integers(-6, 7, 4096)
    X[:, 1] = rng.standard_normal(4096) * 10.0 ** rng.integers(-6, 7, 4096)
    special = [(1.0, 0.0), (-2.5, 0.0), (0.0, 0.0), (3.0, np.inf), (-3.0, -np.inf), (np.inf, 2.0), (np.nan, 1.0), (1.0, np.nan),
               (np.inf, np.inf), (0.0, 5.0), (-0.0, 5.0), (1.0, -0.0)]
    for i, (a, b) in enumerate(special):
        X[i, 0], X[i, 1] = a, b
    got = rv.SymbolicRegressor("x0 / x1").predict(X)
    with np.errstate(all="ignore"):
        want = X[:, 0] / X[:, 1]
    n = len(special)
    assert np.array_equal(np.isnan(got[:n]), np.isnan(want[:n]))
    assert np.array_equal(got[:n][~np.isnan(want[:n])], want[:n][~np.isnan(want[:n])])          # +-inf, +-0 exactly
    assert np.array_equal(np.signbit(got[:n][want[:n] == 0]), np.signbit(want[:n][want[:n] == 0]))
    np.testing.assert_allclose(got[n:], want[n:], rtol=4e-16, atol=0)
    # operands whose reciprocal or quotient leaves the exponent range take the IEEE sequence (m_divx): generation-2 features are
    # unscaled, nothing bounds them -- finite quotients stay finite, overflow is inf, underflow is a (sub)normal or zero
    ext = [(1e300, 1e-300), (1e-300, 1e300), (3.0, 1e-310), (1e308, 1e308), (2e-308, 4.0), (1e200, 1e-200), (-7e305, 1e-10),
           (1e-320, 1e-320), (5e150, 2e160), (1.0, 1e308), (1e308, 0.5)]
    X2 = np.zeros((len(ext), 18))
    X2[:, 0] = [a for a, _ in ext]; X2[:, 1] = [b for _, b in ext]
    got2 = rv.SymbolicRegressor("x0 / x1").predict(X2)
    with np.errstate(all="ignore"):
        want2 = X2[:, 0] / X2[:, 1]
    np.testing.assert_allclose(got2, want2, rtol=4e-16, atol=0)


def test_solve_catenary_and_tension(rv, golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_solve_catenary.npz"))
    C = rv.solve_catenary(g["l"], g["dH"], float(g["L"]))
    assert np.array_equal(np.isnan(C), np.isnan(g["C"]))
    np.testing.assert_allclose(C, g["C"], rtol=0, atol=1e-11, equal_nan=True)
    C2, T = rv.cable_tension(g["l"], g["dH"], float(g["L"]), float(g["w_wet"]))
    np.testing.assert_allclose(T, g["T"], rtol=1e-10)
    C3 = rv.solve_catenary(g["l_in"], g["dH_in"], 3.0)
    assert np.array_equal(np.isnan(C3), np.isnan(g["C_in"]))
    np.testing.assert_allclose(C3, g["C_in"], rtol=0, atol=1e-11, equal_nan=True)
    # SURVEY KATs incl. both failure modes (taut, root above the bracket)
    k = rv.solve_catenary([1.0, 1.41421356, 2.0, 2.5, 2.9, 0.5, 3.5], [0, -1, 0.5, -0.3, 0.1, 0, 0], 3.0)
    np.testing.assert_allclose(k[:5], [5.676892760096155, 3.0791940475045547, 1.5916068034624558,
                                       0.8396630142778874, 0.309507897379277], rtol=1e-11)
    assert np.isnan(k[5]) and np.isnan(k[6])
    assert rv.solve_catenary(np.zeros((0,)), np.zeros((0,)), 3.0).shape == (0,)      # empty input
    assert rv.solve_catenary(np.full((2, 3), 1.0), 0.0, 3.0).shape == (2, 3)         # broadcasting


def test_solve_catenary_large_random_vs_brentq(rv, orc):
    rng = np.random.default_rng(11)
    l = rng.uniform(1e-3, 3.5, 4000); dH = rng.uniform(-3.2, 3.2, 4000)
    ref = orc.solve_catenary_ref(l, dH, 3.0)
    got = rv.solve_catenary(l, dH, 3.0)
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11, equal_nan=True)


def test_rodrigues(rv, golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_rodrigues.npz"))
    out = rv.default_engine().rodrigues(g["v"], g["axis"], g["angle"])
    np.testing.assert_allclose(out, g["out"], rtol=1e-13, atol=1e-15)
    one = rv.rodrigues_rotation(np.array([1., 1., -1.]), np.array([1., -1., 0.]) / np.sqrt(2), np.radians(-10))
    np.testing.assert_allclose(one, [0.86201995, 0.86201995, -1.23038336], atol=5e-9)
    assert one.shape == (3,)


def test_transform_catenary_fused_and_generic(rv, orc, golden_dir):
    g = np.load(os.path.join(golden_dir, "kat_transform_catenary.npz"))
    M = int(g["M"])
    cat = rv.Catenary(float(g["L"]), "ENU", n_points=M)
    ocat = orc.Catenary(float(g["L"]), "ENU", n_points=M)
    for i in range(len(g["theta"])):
        n0, n1 = g["npts"][i]
        for fn in (cat, ocat):       # fused kernel path / generic-callable path
            r = rv.transform_catenary(g["A"][i], g["B"][i], fn, g["theta"][i], g["gamma"][i])
            for out, key, n in zip(r, ("original", "theta_rotated", "theta_aligned", "final"), (n0, n1, n1, n1)):
                assert out.shape == (n, 3)
                np.testing.assert_allclose(out, g[key][i][:n], rtol=1e-10, atol=1e-12)
    out, npts, z = rv.transform_catenary_batch(g["A"], g["B"], g["theta"], g["gamma"], cat)
    assert np.array_equal(npts, g["npts"])
    zref = np.array([np.min(g["final"][i][:g["npts"][i][1], 2]) for i in range(len(z))])
    np.testing.assert_allclose(z, zref, rtol=1e-10, atol=1e-12)


def test_compute_catenary_3d_and_the_fully_pinned_augmented_path(rv, orc, golden_dir):
    """compute_catenary_3D (models/catenary_3d.py:5-39) on the GPU against the outputs of the reference's own function, one
    pair at a time and batched; then transform_catenary with Catenary3D as catenary_fn against main_fun.transform_catenary
    run on that function -- every number on this path is pinned to reference code.
    Tolerance: the reference's sag is a cosh(h / a) - a cosh(x / a) with the a its (diverging) fixed point ends on, 1e6..1e30
    for most pairs, i.e. rounding noise in multiples of ulp(a); two libms agree on it to a few ulp of a, not better."""
    eps = np.finfo(float).eps
    g = np.load(os.path.join(golden_dir, "kat_catenary_3d.npz"))
    M = int(g["M"])
    for L in np.unique(g["rope"]):
        m = g["rope"] == L
        got = rv.compute_catenary_3D(g["p0"][m], g["p1"][m], float(L), M)
        _, a = rv.Catenary3D(float(L), M).batch(g["p0"][m], g["p1"][m])
        assert got.shape == (int(m.sum()), M, 3)
        tol = 1e-13 + 8 * eps * np.where(np.isfinite(a), a, 0.0)
        want = g["points"][m]
        assert np.array_equal(np.isnan(got), np.isnan(want))       # a pair far shorter than the rope: a overflows, z = inf - inf
        assert np.all(np.nan_to_num(np.abs(got - want)).max(axis=(1, 2)) <= tol)
        np.testing.assert_allclose(got[:, :, :2], want[:, :, :2], rtol=1e-14, atol=1e-15)      # x, y carry no sag
    one = rv.compute_catenary_3D(g["p0"][1], g["p1"][1], float(g["rope"][1]), M)
    assert one.shape == (M, 3)
    pts, a = rv.Catenary3D(3.0, M).batch(g["p0"], g["p1"])
    taut = np.linalg.norm(g["p1"] - g["p0"], axis=1) >= 3.0
    assert np.array_equal(np.isnan(a), taut) and taut.sum() >= 2
    with pytest.raises(ValueError):
        rv.compute_catenary_3D(g["p0"][0], g["p1"][0], 3.0, 1)
    t = np.load(os.path.join(golden_dir, "kat_transform_catenary_3d.npz"))
    cat = rv.Catenary3D(float(t["L"]), int(t["M"]))
    for i in range(len(t["theta"])):
        r = rv.transform_catenary(t["A"][i], t["B"][i], cat, t["theta"][i], t["gamma"][i])
        a0 = cat(t["A"][i], t["B"][i])[0]
        a1 = cat(t["A"][i], r[1][-1])[0]                       # the theta-rotated end point's catenary
        tol = 1e-12 + 16 * eps * max([0.0] + [v for v in (a0, a1) if np.isfinite(v)])
        for out, key in zip(r, ("original", "theta_rotated", "theta_aligned", "final")):
            assert out.shape == t[key][i].shape and np.array_equal(np.isnan(out), np.isnan(t[key][i]))
            err = np.nan_to_num(np.abs(out - t[key][i])).max()
            assert err <= tol, (i, key, err, tol)


def test_catenary_callable_contract(rv, orc):
    a = np.array([0.1, -0.2, 0.3]); b = np.array([1.2, 0.7, -0.4])
    for frame in ("ENU", "NED"):
        got = rv.Catenary(3.0, frame, n_points=24)(a, b)
        want = orc.Catenary(3.0, frame, n_points=24)(a, b)
        assert len(got) == 4 and got[3].shape == (24, 3)
        np.testing.assert_allclose(got[3], want[3], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(got[:3], want[:3], rtol=1e-10)
    assert rv.Catenary(3.0)(a, a + np.array([3.0, 0.0, 0.0]))[3] is None       # taut -> None
    assert rv.Catenary(3.0)(a, a + np.array([0.3, 0.0, 0.0]))[3] is None       # root outside bracket
    with pytest.raises(ValueError):
        rv.Catenary(3.0, "XYZ")


def test_replay_integrators(rv, golden_dir, equations):
    from conftest import chosen_row
    g = np.load(os.path.join(golden_dir, "kat_replay.npz"))
    th0, ga0 = float(g["theta0"]), float(g["gamma0"])
    mt = rv.SymbolicRegressor(chosen_row(equations, "dtheta_dt")["sympy_format"])
    mg = rv.SymbolicRegressor(chosen_row(equations, "dgamma_dt")["sympy_format"])
    np.testing.assert_allclose(rv.rk4_integration(mt, g["Xs"], g["time"], th0), g["rk4_theta"], rtol=1e-11)
    np.testing.assert_allclose(rv.rk4_integration(mg, g["Xs"], g["time"], ga0), g["rk4_gamma"], rtol=1e-11)
    th, ga = rv.integrate_theta_gamma(mt, mg, g["Xs"], g["time"], th0, ga0)
    np.testing.assert_allclose(np.stack([th, ga]), g["euler"], rtol=1e-11)
    mt2 = rv.SymbolicRegressor(equations["dtheta_dt"]["rows"][-1]["sympy_format"])
    mg2 = rv.SymbolicRegressor(equations["dgamma_dt"]["rows"][-1]["sympy_format"])
    th, ga = rv.rk4_theta_gamma(mt2, mg2, g["Xs"], g["time"], th0, ga0)
    np.testing.assert_allclose(th, g["rk4_theta_last"], rtol=1e-11)
    np.testing.assert_allclose(ga, g["rk4_gamma_last"], rtol=1e-11)
    th, ga = rv.integrate_theta_gamma(mt2, mg2, g["Xs"], g["time"], th0, ga0)
    np.testing.assert_allclose(np.stack([th, ga]), g["euler_last"], rtol=1e-11)
    one = rv.rk4_integration(mt, g["Xs"][:1], g["time"][:1], th0)                # T = 1: just y0
    assert one.shape == (1,) and one[0] == th0
    np.testing.assert_allclose(mt.predict(g["Xs"]), mt._pair(mt).predict(g["Xs"], 0))


def test_second_order_replays(rv, orc, golden_dir, equations):
    """Double-Euler (test_cluster.py:110-129) and trapezoid+cumsum (dd_cluster.py:221-226) replays of
    second-derivative models, against the oracle's statement-by-statement restatement."""
    g = np.load(os.path.join(golden_dir, "kat_replay.npz"))
    et = equations["dtheta_dt"]["rows"][-1]["sympy_format"]; eg = equations["dgamma_dt"]["rows"][-1]["sympy_format"]
    mt, mg = rv.SymbolicRegressor(et), rv.SymbolicRegressor(eg)
    ddt = orc.SymbolicModel(et).predict(g["Xs"]); ddg = orc.SymbolicModel(eg).predict(g["Xs"])
    th, ga = rv.integrate_second_order(mt, mg, g["Xs"], g["time"], -0.03, -0.05, "double_euler")
    wt, wg = orc.double_euler_replay(ddt, ddg, g["time"], -0.03, -0.05)
    np.testing.assert_allclose(th, wt, rtol=1e-11, atol=1e-14); np.testing.assert_allclose(ga, wg, rtol=1e-11, atol=1e-14)
    th, ga = rv.integrate_second_order(mt, mg, g["Xs"], g["time"], -0.03, -0.05, "trapezoid")
    wt, wg = orc.trapezoid_replay(ddt, ddg, g["time"], -0.03, -0.05)
    np.testing.assert_allclose(th, wt, rtol=1e-11, atol=1e-14); np.testing.assert_allclose(ga, wg, rtol=1e-11, atol=1e-14)
    with pytest.raises(ValueError):
        rv.integrate_second_order(mt, mg, g["Xs"], g["time"], 0.0, 0.0, "simpson")


def test_kabsch_velocity_transform(rv, orc, golden_dir):
    """Batched Kabsch (3x3 SVD + reflection fix) against the reference's own function on cable-like
    (nearly planar), exactly planar and generic marker sets, plus the per-frame gates."""
    g = np.load(os.path.join(golden_dir, "kat_kabsch.npz"))
    v, R = rv.kabsch_velocity_transform(g["P"], g["Q"], g["v"])
    np.testing.assert_allclose(R, g["R"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(v, g["v_out"], rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(rv.compute_rotation_kabsch(g["P"][3], g["Q"][3]), g["R"][3], atol=1e-10)
    # gates: NaN marker, no motion (batch variant only), reflection case
    P = g["P"][:4].copy(); Q = g["Q"][:4].copy(); vv = g["v"][:4].copy()
    P[1, 2, 0] = np.nan
    Q[2] = P[2]
    Q[3] = P[3] * np.array([1.0, 1.0, -1.0])          # mirrored set: best PROPER rotation, det = +1
    got, Rg = rv.kabsch_velocity_transform(P, Q, vv, batch_gates=True)
    want, Rw = orc.kabsch_velocity_transform(P, Q, vv, batch_gates=True)
    assert np.isnan(got[1]).all() and np.isnan(got[2]).all() and np.isnan(want[2]).all()
    np.testing.assert_allclose(got[[0, 3]], want[[0, 3]], rtol=1e-9, atol=1e-8)
    assert np.linalg.det(Rg[3]) == pytest.approx(1.0, abs=1e-10)
    got2, _ = rv.kabsch_velocity_transform(P, Q, vv, batch_gates=False)
    want2, _ = orc.kabsch_velocity_transform(P, Q, vv, batch_gates=False)
    assert np.isnan(got2[1]).all() and np.isfinite(got2[2]).all()
    np.testing.assert_allclose(got2[[0, 2, 3]], want2[[0, 2, 3]], rtol=1e-9, atol=1e-8)
    with pytest.raises(ValueError):
        rv.kabsch_velocity_transform(P, Q[:2], vv)


def test_extract_features_gpu(rv, golden_dir):
    """simply.py:15-41 / main_fun.py:167-193 on the GPU (np.gradient with non-uniform time) against the
    reference's own outputs."""
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "kat_features.npz"))
    df = pd.DataFrame(g["frame"], columns=[str(c) for c in g["columns"]])
    np.testing.assert_allclose(rv.extract_features(df), g["X18"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(rv.extract_features(df, with_prev=False), g["X16"], rtol=1e-12, atol=1e-13)
    with pytest.raises(rv.RovmpcError):
        rv.extract_features_arrays(np.zeros((1, 3)), np.ones((1, 3)), np.ones((1, 3)), [0.0], [0.0], [0.0])


def test_velocity_transform(rv):
    rng = np.random.default_rng(3)
    R = rand_rtab(50); v = rng.standard_normal((50, 3))
    np.testing.assert_allclose(rv.velocity_transform(R, v), np.einsum("tij,tj->ti", R, v), rtol=1e-14, atol=1e-15)


# ---------------------------------------------------------------------------------------------
# the hot path vs the oracle
# ---------------------------------------------------------------------------------------------

def run_both(rv, orc, cfg, model=None, seed=20250523, Rtab=None):
    model = model or rv.default_model()
    state, U = rv.synthetic_problem(cfg.K, cfg.N, seed, cfg.np_dtype)
    with rv.Engine(cfg, model) as e:
        if Rtab is not None:
            e.set_rotation_table(Rtab)
        J, traj = e.rollout_costs(state, U, return_traj=True)
        res = e.step(state, U)
    Jo, trajo, aux = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, model), orc.MPCState.from_array(state),
                                     U.astype(np.float64), Rtab)
    return (J, traj, res), (Jo, trajo, aux), (state, U)


@pytest.mark.parametrize("vt_mode", [0, 1, 2])
@pytest.mark.parametrize("prev_mode,integrator", [(0, 0), (1, 0), (0, 1)])
@pytest.mark.parametrize("force_interp", [False, True])
def test_rollout_c1_all_modes(rv, orc, vt_mode, prev_mode, integrator, force_interp):
    """BASELINE config 1 (N=20, K=64) in every mode of the kernel, compiled-in and interpreted."""
    cfg = rv.MPCConfig(N=20, K=64, vt_mode=vt_mode, prev_mode=prev_mode, integrator=integrator,
                       force_interpreter=force_interp)
    Rtab = rand_rtab(20) if vt_mode == 2 else None
    (J, traj, res), (Jo, trajo, aux), (state, U) = run_both(rv, orc, cfg, Rtab=Rtab)
    np.testing.assert_allclose(traj, trajo, rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(J, Jo, rtol=RTOL)
    k = int(np.argmin(Jo))
    assert res.index == k
    assert np.array_equal(res.u, U[k, 0])
    assert res.cost == pytest.approx(Jo[k], rel=RTOL)
    np.testing.assert_allclose(res.traj, trajo[k], rtol=RTOL, atol=1e-13)


def test_rollout_c1_scalar_reference_style_oracle(rv, orc):
    """Same, against the reference-style scalar flavour (per-row predict, scipy brentq, per-point
    Rodrigues loops) on a few candidates."""
    cfg = rv.MPCConfig(N=20, K=8)
    state, U = rv.synthetic_problem(cfg.K, cfg.N)
    with rv.Engine(cfg) as e:
        J, traj = e.rollout_costs(state, U, return_traj=True)
    Jo, trajo, _ = orc.rollout_scalar(oracle_cfg(orc, cfg), oracle_model(orc, rv.default_model()),
                                      orc.MPCState.from_array(state), U)
    np.testing.assert_allclose(traj, trajo, rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(J, Jo, rtol=RTOL)


@pytest.mark.parametrize("ck", [0, 1, 2, 8, 16])
def test_step_c2_full_size(rv, orc, ck):
    """BASELINE config 2: N=20, K=4096, fp64 -- chosen u identical, trajectory within 1e-5 (held to 1e-9)."""
    cfg = rv.MPCConfig(N=20, K=4096, candidates_per_block=ck)
    (J, traj, res), (Jo, trajo, aux), (state, U) = run_both(rv, orc, cfg)
    k = int(np.argmin(Jo))
    assert res.index == k and np.array_equal(res.u, U[k, 0])
    np.testing.assert_allclose(res.traj, trajo[k], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(J, Jo, rtol=RTOL)
    np.testing.assert_allclose(traj, trajo, rtol=RTOL, atol=1e-13)


@pytest.mark.parametrize("jit", [True, False])
@pytest.mark.parametrize("ct,cg", [(30, 27), (12, 16), (8, 7), (1, 1), (29, 23)])
def test_rollout_other_pareto_rows(rv, orc, ct, cg, jit):
    """Other rows of the reference's Pareto fronts (abs, tanh, square, nested sin): the hiprtc-
    specialised kernel and the bytecode interpreter."""
    model = rv.default_model(ct, cg)
    cfg = rv.MPCConfig(N=12, K=96, jit=jit)
    with rv.Engine(cfg, model) as e:
        assert e.model_path == ("jit" if jit else "interpreter")
    (J, traj, res), (Jo, trajo, aux), _ = run_both(rv, orc, cfg, model)
    np.testing.assert_allclose(traj, trajo, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(J, Jo, rtol=1e-8)
    assert res.index == int(np.argmin(Jo))


@pytest.mark.parametrize("vt_mode,prev_mode,integrator,dtype", [(0, 0, 0, "f64"), (2, 1, 0, "f64"), (1, 0, 1, "f64"), (1, 0, 0, "f32")])
def test_jit_specialisation_modes(rv, orc, vt_mode, prev_mode, integrator, dtype):
    """hiprtc-specialised kernel for a model with the generation-2 operator mix, in the other kernel
    modes, and that the default rows loaded through the JIT route (compiled-in substitution switched
    off with cfg.no_builtin) agree with the compiled-in kernel."""
    mean, scale = rv.default_model().mean, rv.default_model().scale
    model = rv.DynamicsModel(mean, scale, "0.05*(sin(x17) - tanh(x3*1.6) - x16) + 0.01*Abs(x11)*x6/(1.0 + x12**2) - 0.002*exp(-x0**2)",
                             "x15 - x17 + 0.008*(x3 - sin(x15)) + 0.001*x13*cos(x9)")
    cfg = rv.MPCConfig(N=10, K=80, vt_mode=vt_mode, prev_mode=prev_mode, integrator=integrator, dtype=dtype)
    Rtab = rand_rtab(10) if vt_mode == 2 else None
    with rv.Engine(cfg, model) as e:
        assert e.model_path == "jit"
    (J, traj, res), (Jo, trajo, aux), _ = run_both(rv, orc, cfg, model, Rtab=Rtab)
    tol = 1e-8 if dtype == "f64" else 2e-3
    np.testing.assert_allclose(traj, trajo, rtol=tol, atol=1e-12 if dtype == "f64" else 1e-5)
    np.testing.assert_allclose(J, Jo, rtol=tol)
    if dtype == "f64":
        assert res.index == int(np.argmin(Jo))
        # the reference's own rows through the hiprtc route (no_builtin) against the compiled-in kernel
        cfg2 = rv.MPCConfig(N=20, K=256, vt_mode=vt_mode, prev_mode=prev_mode, integrator=integrator)
        cfgj = rv.MPCConfig(N=20, K=256, vt_mode=vt_mode, prev_mode=prev_mode, integrator=integrator, no_builtin=True)
        state, U = rv.synthetic_problem(256, 20)
        with rv.Engine(cfgj) as ej, rv.Engine(cfg2) as eb:
            if Rtab is not None:
                R20 = rand_rtab(20); ej.set_rotation_table(R20); eb.set_rotation_table(R20)
            assert eb.model_path == "builtin" and ej.model_path == "jit"
            np.testing.assert_allclose(ej.rollout_costs(state, U), eb.rollout_costs(state, U), rtol=1e-11)


@pytest.mark.parametrize("jit", [True, False])
@pytest.mark.parametrize("vt_mode", [0, 1])
def test_generation2_feature_map(rv, orc, jit, vt_mode):
    """The model simulate_rk4_theta_gamma.py actually loads: 17 unscaled features incl. cos(theta),
    sin(gamma) and the unclipped angle_proj; rows 13 / 20 of outputs/differential_training_new_feature/."""
    model = rv.generation2_model()
    cfg = rv.MPCConfig(N=10, K=72, feature_map=rv.FEATURES_GEN2, jit=jit, vt_mode=vt_mode)
    with rv.Engine(cfg, model) as e:
        assert e.model_path == ("jit" if jit else "interpreter")
    (J, traj, res), (Jo, trajo, _), (state, U) = run_both(rv, orc, cfg, model)
    np.testing.assert_allclose(traj, trajo, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(J, Jo, rtol=1e-8)
    assert res.index == int(np.argmin(Jo))
    with pytest.raises(rv.RovmpcError):                       # 17-slot model on the 18-slot map
        with rv.Engine(rv.MPCConfig(N=4, K=8), model) as e:
            e.step(state, U[:8, :4])


@pytest.mark.parametrize("jit", [True, False])
@pytest.mark.parametrize("vt_mode,integrator", [(0, 0), (1, 0), (2, 0), (1, 1), (0, 1)])
def test_generation3_second_order_rollout(rv, orc, jit, vt_mode, integrator):
    """Second-order models (dd_cluster.py; rows 6 / 5 of outputs/dd_C6_all_50_s_20250511_013928/) on the 14 named
    features of features_dd: state (theta, gamma, dtheta, dgamma), RK4 and the reference's double Euler."""
    model = rv.generation3_model()
    cfg = rv.MPCConfig(N=12, K=80, feature_map=rv.FEATURES_GEN3, jit=jit, vt_mode=vt_mode, integrator=integrator)
    state, U = rv.synthetic_problem(cfg.K, cfg.N, 20250523)
    state[14], state[15] = 0.013, -0.021                       # dtheta_0, dgamma_0
    state[9:12] = (5.0, -3.0, 2.0)
    Rtab = rand_rtab(cfg.N) if vt_mode == 2 else None
    with rv.Engine(cfg, model) as e:
        assert e.model_path == ("jit" if jit else "interpreter")
        if Rtab is not None:
            e.set_rotation_table(Rtab)
        J, traj = e.rollout_costs(state, U, return_traj=True)
        res = e.step(state, U)
    om = orc.DynamicsModel(model.mean, model.scale, orc.SymbolicModel(model.expr_theta, 14, model.variable_names),
                           orc.SymbolicModel(model.expr_gamma, 14, model.variable_names))
    Jo, trajo, _ = orc.rollout_vec_dd(oracle_cfg(orc, cfg), om, orc.MPCState.from_array(state), U, Rtab)
    np.testing.assert_allclose(traj, trajo, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(J, Jo, rtol=1e-8)
    k = int(np.argmin(Jo))
    assert res.index == k and np.array_equal(res.u, U[k, 0])
    np.testing.assert_allclose(res.traj, trajo[k], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("ct,cg", [(9, 8), (14, 9), (22, 18), (4, 16)])
def test_generation3_other_rows(rv, orc, ct, cg):
    """Other Pareto rows of the second-order run (division by V_x, nested sin, the 'gama' variable)."""
    model = rv.generation3_model(ct, cg)
    cfg = rv.MPCConfig(N=8, K=48, feature_map=rv.FEATURES_GEN3)
    state, U = rv.synthetic_problem(cfg.K, cfg.N, 3)
    state[14], state[15] = -0.02, 0.03
    with rv.Engine(cfg, model) as e:
        J, traj = e.rollout_costs(state, U, return_traj=True)
    om = orc.DynamicsModel(model.mean, model.scale, orc.SymbolicModel(model.expr_theta, 14, model.variable_names),
                           orc.SymbolicModel(model.expr_gamma, 14, model.variable_names))
    Jo, trajo, _ = orc.rollout_vec_dd(oracle_cfg(orc, cfg), om, orc.MPCState.from_array(state), U)
    ok = np.isfinite(Jo)
    assert ok.sum() > cfg.K // 2
    np.testing.assert_allclose(traj[ok], trajo[ok], rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(J[ok], Jo[ok], rtol=1e-7)


def test_generation3_predict_matches_reference_rows(rv, golden_dir):
    """Every row of the second-order Pareto fronts through rovmpc_predict vs sympy-lambdified reference text."""
    import json
    g = np.load(os.path.join(golden_dir, "kat_dynamics_gen3.npz"))
    eq = json.load(open(os.path.join(golden_dir, "equations_gen3.json")))
    sc = json.load(open(os.path.join(golden_dir, "scaler_gen3.json")))
    for which, key in (("ddtheta", "out_theta"), ("ddgamma", "out_gamma")):
        for i, r in enumerate(eq[which]["rows"]):
            m = rv.DynamicsModel(np.zeros(14), np.ones(14), r["sympy_format"], "0.0*theta", variable_names=eq["variable_names"])
            with rv.Engine(rv.MPCConfig(N=2, K=2, feature_map=rv.FEATURES_GEN3), m) as e:
                out_t = e.predict(g["X"], 0)
            ref = g[key][i]
            fin = np.isfinite(ref)
            np.testing.assert_allclose(out_t[fin], ref[fin], rtol=1e-10, atol=1e-13)
    assert len(sc["mean"]) == 14


def test_features_dd_equal_the_reference(rv, orc, golden_dir):
    """rovmpc_features_dd vs main_fun.features_dd's own output on the golden log (Savitzky-Golay 11/3 with
    scipy's edge polynomial, three chained np.gradient passes), then other windows vs the oracle."""
    d = np.load(os.path.join(golden_dir, "kat_features_dd.npz"))
    with rv.Engine(rv.MPCConfig(N=2, K=2)) as e:
        F, Y = e.features_dd(d["P0"], d["P1"], d["V"], d["time"], d["theta"], d["gamma"])
        np.testing.assert_allclose(F, d["features"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(Y, d["targets"], rtol=1e-8, atol=1e-9)
        # the shortest admissible log (T = window) and a uniform time base
        T = 11
        t = np.arange(T) * 0.05
        F2, Y2 = e.features_dd(d["P0"][:T], d["P1"][:T], d["V"][:T], t, d["theta"][:T], d["gamma"][:T])
        Fo, Yo = orc.features_dd(d["P0"][:T], d["P1"][:T], d["V"][:T], t, d["theta"][:T], d["gamma"][:T])
        np.testing.assert_allclose(F2, Fo, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(Y2, Yo, rtol=1e-8, atol=1e-9)
        with pytest.raises(ValueError):
            e.features_dd(d["P0"][:8], d["P1"][:8], d["V"][:8], t[:8], d["theta"][:8], d["gamma"][:8])
        # other window / order against scipy directly
        from scipy.signal import savgol_filter
        F3, _ = e.features_dd(d["P0"], d["P1"], d["V"], d["time"], d["theta"], d["gamma"], window=21, polyorder=5)
        np.testing.assert_allclose(F3[:, 0], savgol_filter(d["theta"], 21, 5), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(F3[:, 1], savgol_filter(d["gamma"], 21, 5), rtol=1e-9, atol=1e-12)


def test_smoothing_helpers_equal_the_reference(rv, golden_dir):
    """preprocess_signals (gaussian_filter1d sigma 2, 'reflect') and compute_derivatives vs main_fun's own outputs."""
    import pandas as pd
    d = np.load(os.path.join(golden_dir, "kat_features_dd.npz"))
    g = np.load(os.path.join(golden_dir, "kat_smoothing.npz"))
    df = pd.DataFrame({"Time": d["time"], "Theta": d["theta"], "Gamma": d["gamma"]})
    t, th, ga = rv.preprocess_signals(df, sigma=2)
    np.testing.assert_allclose(th, g["theta_gauss2"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(ga, g["gamma_gauss2"], rtol=1e-12, atol=1e-15)
    t, th, ga = rv.preprocess_signals(df.iloc[:6], sigma=3.5)      # kernel radius 14 > 6 rows: repeated reflection
    np.testing.assert_allclose(th, g["theta_gauss35_first6"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(ga, g["gamma_gauss35_first6"], rtol=1e-12, atol=1e-15)
    ddt, ddg = rv.compute_derivatives(df)
    np.testing.assert_allclose(ddt, g["ddtheta"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(ddg, g["ddgamma"], rtol=1e-8, atol=1e-9)


def test_rollout_c3_fp32(rv, orc):
    """BASELINE config 3: N=50, K=16384 in fp32, checked against the fp64 oracle with the rule of
    SURVEY section 8(d): same k*, or |J32 - J64| / J64 < 1e-4 at both minimisers."""
    cfg = rv.MPCConfig(N=50, K=16384, dtype="f32")
    (J, traj, res), (Jo, trajo, aux), (state, U) = run_both(rv, orc, cfg)
    k = int(np.argmin(Jo))
    ok = res.index == k or (abs(J[k] - Jo[k]) / Jo[k] < 1e-4 and abs(J[res.index] - Jo[res.index]) / Jo[res.index] < 1e-4
                            and abs(Jo[res.index] - Jo[k]) / Jo[k] < 1e-4)
    assert ok
    rel = np.abs(J - Jo) / np.abs(Jo)
    assert np.median(rel) < 1e-5 and rel.max() < 2e-3
    terr = np.abs(traj - trajo).max() / np.abs(trajo).max()
    assert terr < 1e-4
    # and the same problem in fp64 on the GPU is exact
    cfg64 = rv.MPCConfig(N=50, K=16384, dtype="f64")
    with rv.Engine(cfg64) as e:
        r64 = e.step(state, U.astype(np.float64))
    assert r64.index == k


def test_long_horizon_multi_round_grid_shares_the_gamma_table(rv, orc):
    """N = 50, K = 16384 (config 3's size) in fp64: the grid runs in two rounds, the second round's workgroups load the gamma
    table workgroup 0 published instead of integrating gamma again, the geometry waves chase the theta wave and the sines are
    shared between waves -- all of it must leave the costs where the oracle has them, and repeated launches (fresh epochs,
    another state in between) must reproduce themselves bit for bit."""
    import torch
    N, K = 50, 16384
    cfg = rv.MPCConfig(N=N, K=K)
    state, U = rv.synthetic_problem(K, N, seed=21)
    state2 = state.copy(); state2[13] += 0.01; state2[15] -= 0.005           # another gamma path
    with rv.Engine(cfg) as e:
        J1 = e.rollout_costs(state, U)
        J2 = e.rollout_costs(state2, U)
        J1b = e.rollout_costs(state, U)
        r = e.step(state, U)
    assert np.array_equal(J1, J1b) and not np.array_equal(J1, J2)
    om = oracle_model(orc, rv.default_model())
    for st_, J in ((state, J1), (state2, J2)):
        Jo, trajo, _ = orc.rollout_vec(oracle_cfg(orc, cfg), om, orc.MPCState.from_array(st_), U)
        np.testing.assert_allclose(J, Jo, rtol=1e-9)
    Jo, trajo, _ = orc.rollout_vec(oracle_cfg(orc, cfg), om, orc.MPCState.from_array(state), U)
    assert r.index == int(np.argmin(Jo))
    np.testing.assert_allclose(r.traj, trajo[r.index], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("K,N,ck", [(1, 1, 0), (67, 7, 16), (130, 3, 64), (5, 50, 4), (1000, 2, 0)])
def test_ragged_and_tiny_shapes(rv, orc, K, N, ck):
    cfg = rv.MPCConfig(N=N, K=K, candidates_per_block=ck, n_shape_pts=5)
    (J, traj, res), (Jo, trajo, aux), _ = run_both(rv, orc, cfg)
    np.testing.assert_allclose(J, Jo, rtol=RTOL)
    np.testing.assert_allclose(traj, trajo, rtol=RTOL, atol=1e-13)
    assert res.index == int(np.argmin(Jo))


@pytest.mark.parametrize("N,K,ck,vt,prev,integ", [(36, 64, 0, 1, 0, 0), (40, 40, 8, 1, 1, 0), (21, 48, 16, 0, 0, 0),
                                                   (24, 32, 0, 2, 0, 1), (33, 200, 0, 1, 0, 0), (17, 96, 4, 1, 0, 0),
                                                   (64, 64, 0, 1, 0, 0), (20, 8192, 0, 1, 0, 0)])
def test_workgroup_shapes_of_the_compiled_in_path(rv, orc, N, K, ck, vt, prev, integ):
    """Shapes that exercise every dispatch of the compiled-in kernel: early phase-4b batch with one
    and with several post-join rounds, no early batch, several theta waves (32 candidates per
    workgroup), tiny workgroups, Euler and HOLD."""
    cfg = rv.MPCConfig(N=N, K=K, candidates_per_block=ck, vt_mode=vt, prev_mode=prev, integrator=integ, n_shape_pts=7)
    Rtab = rand_rtab(N) if vt == 2 else None
    (J, traj, res), (Jo, trajo, aux), _ = run_both(rv, orc, cfg, Rtab=Rtab)
    np.testing.assert_allclose(traj, trajo, rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(J, Jo, rtol=RTOL)
    assert res.index == int(np.argmin(Jo))


@pytest.mark.parametrize("N,K,ck,nt,dtype", [(20, 600, 16, 256, "f64"), (20, 600, 32, 512, "f64"), (20, 300, 16, 192, "f64"),
                                             (20, 300, 8, 128, "f64"), (20, 200, 16, 64, "f64"), (50, 200, 16, 512, "f64"),
                                             (20, 600, 32, 512, "f32"), (20, 600, 64, 512, "f32"), (20, 600, 32, 192, "f64")])
def test_throughput_geometries(rv, orc, N, K, ck, nt, dtype):
    """threads_per_block: the (candidates, threads) per workgroup pairs the library's throughput rule can
    choose for large candidate sets, plus a narrow one-wave workgroup; results must not depend on it."""
    cfg = rv.MPCConfig(N=N, K=K, candidates_per_block=ck, threads_per_block=nt, dtype=dtype)
    (J, traj, res), (Jo, trajo, aux), _ = run_both(rv, orc, cfg)
    if dtype == "f64":
        np.testing.assert_allclose(traj, trajo, rtol=RTOL, atol=1e-13)
        np.testing.assert_allclose(J, Jo, rtol=RTOL)
        assert res.index == int(np.argmin(Jo))
    else:
        assert np.median(np.abs(J - Jo) / np.abs(Jo)) < 1e-5
        assert np.abs(traj - trajo).max() / np.abs(trajo).max() < 1e-4


def test_auto_geometry_large_candidate_set(rv, orc):
    """K = 16384, N = 20, fp64: more than one 16-candidate workgroup per CU, the library switches to its
    throughput geometry; k*, u and the trajectory still equal the oracle's."""
    cfg = rv.MPCConfig(N=20, K=16384)
    (J, traj, res), (Jo, trajo, aux), (state, U) = run_both(rv, orc, cfg)
    k = int(np.argmin(Jo))
    assert res.index == k and np.array_equal(res.u, U[k, 0])
    np.testing.assert_allclose(res.traj, trajo[k], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(J, Jo, rtol=RTOL)


def test_threads_per_block_is_validated(rv):
    for bad in (100, 576, -64):
        with pytest.raises(rv.RovmpcError):
            rv.Engine(rv.MPCConfig(N=4, K=8, threads_per_block=bad))


@pytest.mark.parametrize("vt_mode", [0, 1])
def test_huge_sine_arguments_take_the_checked_path(rv, orc, vt_mode):
    """One candidate with a control of 3e10 mm/s (|x3| ~ 3e8, beyond the two-term argument reduction): the
    workgroup that holds it falls back to the checked sine; its neighbours and every other workgroup agree with
    the oracle as before."""
    cfg = rv.MPCConfig(N=6, K=48, vt_mode=vt_mode)
    state, U = rv.synthetic_problem(cfg.K, cfg.N, 9)
    U[5, 2, 0] = 3e10
    U[37, 0, 0] = -8e9
    model = rv.default_model()
    with rv.Engine(cfg, model) as e:
        J, traj = e.rollout_costs(state, U, return_traj=True)
        res = e.step(state, U)
    Jo, trajo, _ = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, model), orc.MPCState.from_array(state), U)
    np.testing.assert_allclose(traj, trajo, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(J, Jo, rtol=1e-8)
    assert res.index == int(np.argmin(Jo))


def test_a_candidate_does_not_depend_on_its_neighbours(rv):
    """Bit-for-bit: the same control sequence gives the same cost and trajectory whatever shares its workgroup
    (NaN, huge or ordinary neighbours change wave-level decisions -- checked sine, re-anchoring -- not results)."""
    cfg = rv.MPCConfig(N=20, K=64)
    state, U = rv.synthetic_problem(cfg.K, cfg.N, seed=21)
    U[40] = U[5]; U[50] = U[5]
    V = U.copy()
    V[3] = np.nan; V[6, 4, 0] = 5e10; V[41] = 0.0          # neighbours of 5 / 40 / 50 in their workgroups
    with rv.Engine(cfg) as e:
        J, traj = e.rollout_costs(state, U, return_traj=True)
        J2, traj2 = e.rollout_costs(state, V, return_traj=True)
    assert J[5] == J[40] == J[50] == J2[5] == J2[40] == J2[50]
    for k in (40, 50):
        assert np.array_equal(traj[5], traj[k]) and np.array_equal(traj2[5], traj2[k]) and np.array_equal(traj[5], traj2[k])


@pytest.mark.parametrize("N,K,dtype,B", [(20, 4096, "f64", 1), (20, 512, "f64", 8), (50, 16384, "f32", 1)])
def test_literal_horizon_instances_equal_the_run_time_ones(rv, N, K, dtype, B, monkeypatch):
    """The BASELINE horizons (N = 20 in double, N = 50 in single precision, 16 candidates per workgroup) run on kernel
    instances with N and CK as literals; ROVMPC_NO_LITERAL_N=1 keeps the horizon a run-time value.  Same arithmetic, other
    addressing: every record bit for bit, single launches and batched ones."""
    import torch
    cfg = rv.MPCConfig(N=N, K=K, dtype=dtype)
    dev = torch.device("cuda", 0)
    states = np.empty((B, 16)); U = np.empty((B, K, N, 3), dtype=cfg.np_dtype)
    for b in range(B):
        states[b], U[b] = rv.synthetic_problem(K, N, seed=91 + b, dtype=cfg.np_dtype)
    d_s, d_U = torch.tensor(states, device=dev), torch.tensor(U, device=dev)
    out = []
    with rv.Engine(cfg) as e:
        for literal in (True, False):
            if literal:
                monkeypatch.delenv("ROVMPC_NO_LITERAL_N", raising=False)
            else:
                monkeypatch.setenv("ROVMPC_NO_LITERAL_N", "1")
            d_r = torch.zeros((B, e.result_len), dtype=torch.float64, device=dev)
            stream = torch.cuda.current_stream().cuda_stream
            if B == 1:
                e.step_device(d_s.data_ptr(), d_U.data_ptr(), d_r.data_ptr(), stream)
                J = e.rollout_costs(states[0], U[0])
            else:
                e.step_batch_device(B, d_s.data_ptr(), d_U.data_ptr(), d_r.data_ptr(), stream)
                J = None
            torch.cuda.synchronize()
            out.append((d_r.cpu().numpy().copy(), J))
    assert np.array_equal(out[0][0], out[1][0])
    if out[0][1] is not None:
        assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("scale,K", [(20.0, 64), (40.0, 40), (20.0, 4096)])
def test_theta_steps_beyond_the_angle_addition_bound(rv, orc, scale, K):
    """Steps with |theta_{n+1} - theta_n| beyond the bound of the angle addition (fast vehicles): the chain notes them and, one
    step later, replaces the angle-addition sincos of those lanes by a full evaluation and runs the step again -- mixed with
    small steps in the same waves (controls x 20) and in most steps (x 40).  dt = 1/60: the
    reference's gamma equation amplifies rounding by 1e20 over twenty steps of 0.05 s, which would test nothing.
    Against the oracle, and the records of equal candidates bit for bit."""
    cfg = rv.MPCConfig(N=20, K=K, dt=1 / 60)
    state, U = rv.synthetic_problem(cfg.K, cfg.N, seed=33)
    U *= scale
    U[K - 3] = U[1]
    model = rv.default_model()
    with rv.Engine(cfg, model) as e:
        J, traj = e.rollout_costs(state, U, return_traj=True)
        res = e.step(state, U)
    Jo, trajo, _ = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, model), orc.MPCState.from_array(state), U)
    d = np.abs(np.diff(trajo[:, :, 0], axis=1))
    assert (d >= 2.0 ** -6).mean() > 0.1 and (d < 2.0 ** -6).mean() > 0.1
    np.testing.assert_allclose(traj, trajo, rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(J, Jo, rtol=RTOL)
    assert res.index == int(np.argmin(Jo))
    assert J[1] == J[K - 3] and np.array_equal(traj[1], traj[K - 3])


@pytest.mark.parametrize("N,K,dt,vt", [(150, 40, 1 / 240, 1), (100, 33, 1 / 240, 0), (21, 1, 1 / 60, 1), (23, 4097, 1 / 60, 1)])
def test_long_horizons_and_odd_sizes(rv, orc, N, K, dt, vt):
    """Horizons beyond one round of the table passes (3N + 2 > 64 items on the gamma wave), small workgroups forced
    by the LDS budget, a single candidate, one candidate past a multiple of the workgroup size."""
    cfg = rv.MPCConfig(N=N, K=K, dt=dt, vt_mode=vt)
    (J, traj, res), (Jo, trajo, aux), _ = run_both(rv, orc, cfg, seed=N)
    np.testing.assert_allclose(traj, trajo, rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(J, Jo, rtol=RTOL)
    assert res.index == int(np.argmin(Jo))


def test_geometry_edge_cases_in_rollout(rv, orc):
    """Taut cable, root above the bracket (tension fallback + straight-segment shape), NED frame,
    a vertical cable (degenerate xy projection)."""
    model = rv.default_model()
    for P1, frame in (((2.6, 1.4, 0.2), "ENU"), ((0.25, 0.1, 0.05), "ENU"), ((0.24, -0.76, 0.3), "NED"),
                      ((0.0, 0.0, -1.4), "ENU"), ((2.95, 0.0, 0.5), "ENU")):
        cfg = rv.MPCConfig(N=6, K=40, frame=frame, z_floor=-0.5 if frame == "ENU" else 0.5)
        state, U = rv.synthetic_problem(cfg.K, cfg.N, seed=5)
        state[3:6] = P1
        with rv.Engine(cfg, model) as e:
            J, traj = e.rollout_costs(state, U, return_traj=True)
        Jo, trajo, aux = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, model), orc.MPCState.from_array(state), U)
        np.testing.assert_allclose(traj, trajo, rtol=RTOL, atol=1e-13)
        np.testing.assert_allclose(J, Jo, rtol=1e-8)


def test_nan_costs_never_win_and_ties_take_lowest_index(rv, orc):
    cfg = rv.MPCConfig(N=5, K=50)
    state, U = rv.synthetic_problem(cfg.K, cfg.N, seed=9)
    U[3] = np.nan                      # poisoned candidate -> NaN cost -> +inf
    U[17] = U[31] = U[44] = U[10]      # exact ties
    U[10] *= 0.0; U[17] *= 0.0; U[31] *= 0.0; U[44] *= 0.0
    with rv.Engine(cfg) as e:
        J = e.rollout_costs(state, U)
        res = e.step(state, U)
    assert np.isposinf(J[3])
    assert J[10] == J[17] == J[31] == J[44]
    Jo, _, _ = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, rv.default_model()), orc.MPCState.from_array(state), U)
    assert res.index == int(np.argmin(Jo))
    # force the tie to be the minimum: zero control cost dominates with a huge w_u
    cfg2 = rv.MPCConfig(N=5, K=50, w_u=1.0)
    with rv.Engine(cfg2) as e:
        res2 = e.step(state, U)
    assert res2.index == 10


def test_permutation_and_shard_properties_full_size(rv):
    """Size-independent properties at K=4096: J is a per-candidate function (permuting candidates
    permutes J bit-for-bit); the arg-min over shards equals the global arg-min."""
    cfg = rv.MPCConfig(N=20, K=4096)
    state, U = rv.synthetic_problem(cfg.K, cfg.N, seed=1)
    perm = np.random.default_rng(0).permutation(cfg.K)
    with rv.Engine(cfg) as e:
        J = e.rollout_costs(state, U)
        Jp = e.rollout_costs(state, U[perm])
        full = e.step(state, U)
    assert np.array_equal(Jp, J[perm])
    assert full.index == int(np.argmin(J)) and full.cost == J.min()
    with rv.Engine(rv.MPCConfig(N=20, K=1024)) as e4:
        parts = [e4.step(state, U[i * 1024:(i + 1) * 1024]) for i in range(4)]
    best = min(range(4), key=lambda i: (parts[i].cost, parts[i].index + i * 1024))
    assert parts[best].index + best * 1024 == full.index and parts[best].cost == full.cost


def test_device_api_and_sharded_select(rv):
    """Device-pointer entry points with torch tensors; G shards on one GPU + emulated
    all-reduce(min) + select kernel == un-sharded step; kernels agree with the host statement
    (pack_record / select_record) used by the gloo tests."""
    import torch
    from rovmpc import sharded as sh
    cfg = rv.MPCConfig(N=20, K=4096)
    state, U = rv.synthetic_problem(cfg.K, cfg.N, seed=2)
    dev = torch.device("cuda", 0)
    d_state = torch.tensor(state, device=dev); d_U = torch.tensor(U, device=dev)
    with rv.Engine(cfg) as e:
        want = e.step(state, U)
        d_res = torch.empty(e.result_len, dtype=torch.float64, device=dev)
        e.step_device(d_state.data_ptr(), d_U.data_ptr(), d_res.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        r = d_res.cpu().numpy()
    assert r[0] == want.cost and int(r[1]) == want.index and np.array_equal(r[2:5], want.u)
    np.testing.assert_array_equal(r[5:].reshape(-1, 2), want.traj)
    G = 4
    Kl = cfg.K // G
    with rv.Engine(rv.MPCConfig(N=20, K=Kl)) as e:
        R = e.result_len
        slots = torch.empty((G, G, R), dtype=torch.int64, device=dev)
        for g in range(G):
            dUg = d_U[g * Kl:(g + 1) * Kl].contiguous()
            e.step_device_sharded(d_state.data_ptr(), dUg.data_ptr(), g * Kl, g, G, slots[g].data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            local = torch.empty(R, dtype=torch.float64, device=dev)
            e.step_device(d_state.data_ptr(), dUg.data_ptr(), local.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            local[1] += g * Kl
            assert torch.equal(slots[g], sh.pack_record(local, g, G))            # kernel == host statement
        reduced = slots.min(dim=0).values.contiguous()                           # what all-reduce(min) leaves
        out = torch.empty(R, dtype=torch.float64, device=dev)
        e.select_device(reduced.data_ptr(), G, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(out, sh.select_record(reduced))
    o = out.cpu().numpy()
    assert o[0] == want.cost and int(o[1]) == want.index and np.array_equal(o[2:5], want.u)
    np.testing.assert_array_equal(o[5:].reshape(-1, 2), want.traj)


def test_sharded_mpc_single_rank_nccl(rv):
    """ShardedMPC end to end on one rank over the real nccl (RCCL) backend."""
    import torch
    import torch.distributed as dist
    from rovmpc.sharded import ShardedMPC
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg = rv.MPCConfig(N=20, K=2048)
        state, U = rv.synthetic_problem(cfg.K, cfg.N, seed=4)
        dev = torch.device("cuda", 0)
        with rv.Engine(cfg) as e:
            want = e.step(state, U)
            smpc = ShardedMPC(e)
            d_state = torch.tensor(state, device=dev); d_U = torch.tensor(U, device=dev)
            recs = []
            for _ in range(3):
                rec = smpc.step_device(d_state, d_U)       # valid after synchronize() (side stream)
                smpc.synchronize()
                recs.append(rec.clone())
            for rec in recs:
                r = rec.cpu().numpy()
                assert r[0] == want.cost and int(r[1]) == want.index
    finally:
        dist.destroy_process_group()


def test_native_rccl_step_single_rank(rv):
    """rovmpc_step_device_allreduce end to end on a one-rank RCCL communicator created by the
    library itself: same record as the plain step, for several pipelined steps."""
    import torch
    from rovmpc.sharded import NativeShardedMPC
    cfg = rv.MPCConfig(N=20, K=2048)
    dev = torch.device("cuda", 0)
    with rv.Engine(cfg) as e:
        batches = [rv.synthetic_problem(cfg.K, cfg.N, seed=40 + i) for i in range(4)]
        want = [e.step(batches[0][0], U) for _, U in batches]      # one shared state, four candidate batches
        smpc = NativeShardedMPC(e, rank=0, world=1)
        assert e.comm_placement() == ""                                # the collective streams are placed at the first step
        d_state = torch.tensor(batches[0][0], device=dev)
        dU = [torch.tensor(U, device=dev) for _, U in batches]
        outs = [smpc.step_device(d_state, dU[i]) for i in range(4)]    # four steps in flight (SLOTS = 4)
        smpc.synchronize()
        # the placement probe ran against the caller's stream and found a collective stream per communicator that neither
        # shares its hardware queue nor its command-processor pipe (a candidate that does is reported, not starred)
        report = e.comm_placement()
        assert report.startswith("undisturbed") and "stream(s) for" in report and "*" in report, report
        recs = [(i, outs[i].clone()) for i in range(4)]
        outs = [smpc.step_device(d_state, dU[3 - i]) for i in range(4)]  # and the buffers are reusable
        smpc.synchronize()
        recs += [(3 - i, outs[i].clone()) for i in range(4)]
        for i, rec in recs:
            r = rec.cpu().numpy()
            assert r[0] == want[i].cost and int(r[1]) == want[i].index and np.array_equal(r[2:5], want[i].u)
        smpc.close()


def test_builtin_substitution_is_structural(rv):
    """The compiled-in kernel replaces a loaded model only when the model IS the reference's rows (same affine form over
    {x_j, sin x_j}, coefficients to the last bit) -- in either spelling of the CSV -- never for a model that merely agrees
    with them on some sample box."""
    m = rv.default_model()
    cfg = rv.MPCConfig(N=4, K=16)
    spellings = ["((((sin(x17) - sin(x3)) - x16) - x3) * 0.048152514)", m.expr_theta, "0.048152514*(sin(x17) - x3 - sin(x3) - x16) + 0.0*x0"]
    for expr in spellings:
        with rv.Engine(cfg, rv.DynamicsModel(m.mean, m.scale, expr, "x15 - x17")) as e:
            assert e.model_path == "builtin", expr
    others = [m.expr_theta + " + 1e-30*x0",                                  # differs by a term no sample would notice
              m.expr_theta.replace("0.048152514*sin(x17)", "0.048152515*sin(x17)"),
              m.expr_theta + " + 0.0*Abs(x3)",                              # an operator outside the affine algebra
              "0.048152514*(sin(x17 + 1e-12) - x3 - sin(x3) - x16)"]
    for expr in others:
        with rv.Engine(cfg, rv.DynamicsModel(m.mean, m.scale, expr, "x15 - x17")) as e:
            assert e.model_path == "jit", expr
    with rv.Engine(cfg, rv.DynamicsModel(m.mean, m.scale, m.expr_theta, "x15 - 1.0000000001*x17")) as e:
        assert e.model_path == "jit"


def test_code_generator_finds_the_structure_of_loaded_rows(rv, orc):
    """What the compiled-in kernel exploits by hand -- gamma's path is candidate-invariant, dtheta/dt needs no stage loop -- the
    code generator reads off the rows' slot-dependency sets (rovmpc_model_structure) and builds into the hiprtc kernel.  Each
    combination against the oracle: both properties (the reference rows through hiprtc, rows 9 / 3), gamma-invariance only
    (dtheta/dt row 7 reads the stage's gamma), stage-freedom only (rows 5 / 9: dgamma/dt reads the velocity slot), neither;
    RK4 with interpolated and held delay slots, Euler; every velocity-transform mode; fp32; and with the analysis switched off."""
    m = rv.default_model()
    cases = [(13, 3, {"gamma_invariant", "theta_stage_free"}), (9, 3, {"gamma_invariant", "theta_stage_free"}),
             (7, 3, {"gamma_invariant"}), (5, 9, set()), (7, 9, set()), (16, 5, {"gamma_invariant"})]
    N, K = 20, 192
    state, U = rv.synthetic_problem(K, N, seed=11)
    for ct, cg, want in cases:
        model = rv.default_model(ct, cg)
        for vt_mode, prev_mode, integrator in [(1, 0, 0), (1, 1, 0), (1, 0, 1), (0, 0, 0), (2, 0, 0)]:
            cfg = rv.MPCConfig(N=N, K=K, vt_mode=vt_mode, prev_mode=prev_mode, integrator=integrator, no_builtin=True)
            R = rand_rtab(N) if vt_mode == 2 else None
            with rv.Engine(cfg, model) as e:
                if R is not None:
                    e.set_rotation_table(R)
                assert e.model_path == "jit" and e.model_structure == want, (ct, cg, e.model_structure)
                J, traj = e.rollout_costs(state, U, return_traj=True)
            Jo, trajo, _ = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, model), orc.MPCState.from_array(state), U, R)
            np.testing.assert_allclose(J, Jo, rtol=1e-8, err_msg=str((ct, cg, vt_mode, prev_mode, integrator)))
            np.testing.assert_allclose(traj, trajo, rtol=1e-8, atol=1e-12)
    # the model with property (i) only, at the C2 size, picks the oracle's candidate
    model = rv.default_model(7, 3)
    state, U = rv.synthetic_problem(4096, 20, seed=12)
    cfg = rv.MPCConfig(N=20, K=4096)
    with rv.Engine(cfg, model) as e:
        assert e.model_structure == {"gamma_invariant"}
        res = e.step(state, U)
    Jo, trajo, _ = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, model), orc.MPCState.from_array(state), U)
    assert res.index == int(np.argmin(Jo)) and res.cost == pytest.approx(Jo[res.index], rel=1e-9)
    # fp32 and the switch
    cfg = rv.MPCConfig(N=N, K=K, dtype="f32", no_builtin=True)
    state, U = rv.synthetic_problem(K, N, seed=11, dtype=np.float32)
    with rv.Engine(cfg, m) as e:
        assert e.model_structure == {"gamma_invariant", "theta_stage_free"}
        J32 = e.rollout_costs(state, U)
    Jo, _, _ = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, m), orc.MPCState.from_array(state), U.astype(np.float64))
    np.testing.assert_allclose(J32, Jo, rtol=2e-4)
    os.environ["ROVMPC_JIT_NO_STRUCT"] = "1"
    try:
        with rv.Engine(rv.MPCConfig(N=N, K=K, no_builtin=True), m) as e:
            assert e.model_path == "jit" and e.model_structure == set()
            Jg = e.rollout_costs(state.astype(np.float64), U.astype(np.float64))
    finally:
        del os.environ["ROVMPC_JIT_NO_STRUCT"]
    np.testing.assert_allclose(Jg, Jo, rtol=1e-8)


@pytest.mark.parametrize("B,K,N,dtype", [(8, 4096, 20, "f64"), (3, 67, 7, "f64"), (5, 1024, 20, "f32"), (64, 256, 20, "f64")])
def test_batched_problems_equal_single_launches(rv, orc, B, K, N, dtype):
    """rovmpc_step_batch_device: B independent problems (own state, own candidates) in one launch.  Every problem's
    record and costs are bit-equal to its own single launch; problem 0 and B-1 are checked against the oracle."""
    import torch
    cfg = rv.MPCConfig(N=N, K=K, dtype=dtype)
    dev = torch.device("cuda", 0)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    states = np.empty((B, 16)); U = np.empty((B, K, N, 3), dtype=cfg.np_dtype)
    for b in range(B):
        states[b], U[b] = rv.synthetic_problem(K, N, seed=900 + b, dtype=cfg.np_dtype)
        states[b, 12:14] += 0.01 * b                                   # different (theta, gamma) per problem
    d_states = torch.tensor(states, device=dev); d_U = torch.tensor(U, device=dev, dtype=tdt)
    stream = torch.cuda.current_stream().cuda_stream
    with rv.Engine(cfg) as e:
        R = e.result_len
        d_res = torch.full((B, R), float("nan"), dtype=torch.float64, device=dev)
        e.step_batch_device(B, d_states.data_ptr(), d_U.data_ptr(), d_res.data_ptr(), stream)
        torch.cuda.synchronize()
        res = d_res.cpu().numpy()
        import ctypes
        nbytes = B * K * (8 if dtype == "f64" else 4)
        Jall = torch.empty((B, K), dtype=tdt, device=dev)
        # device-to-device copy of the batched cost array (pointer from the ABI)
        ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(Jall.data_ptr()), ctypes.c_void_p(e.batch_costs_ptr()), ctypes.c_size_t(nbytes), 3)
        Jall = Jall.cpu().numpy()
        e.step_batch_device(B, d_states.data_ptr(), d_U.data_ptr(), d_res.data_ptr(), stream)     # and again: epochs advance
        torch.cuda.synchronize()
        assert np.array_equal(d_res.cpu().numpy(), res)
        single = torch.empty(R, dtype=torch.float64, device=dev)
        for b in range(B):
            e.step_device(d_states[b].data_ptr(), d_U[b].data_ptr(), single.data_ptr(), stream)
            torch.cuda.synchronize()
            one = single.cpu().numpy()
            if dtype == "f64":
                assert np.array_equal(one, res[b]), b
            else:
                # fp32: a batched grid may use another workgroup geometry than the single launch, and the two inlined copies
                # of the per-node geometry routine (early batch / main round) contract their fp32 FMAs differently: costs can
                # differ in the last bit between geometries (measured: 8 of 1024, 1 ulp); trajectories never do
                np.testing.assert_allclose(one[0], res[b, 0], rtol=1e-6)
                assert one[1] == res[b, 1] or abs(one[0] - res[b, 0]) <= 1e-6 * abs(one[0])
            if b in (0, B - 1):
                J1 = e.rollout_costs(states[b], U[b])
                if dtype == "f64":
                    assert np.array_equal(J1, Jall[b])
                else:
                    np.testing.assert_allclose(J1, Jall[b], rtol=1e-6)
    model = rv.default_model()
    for b in (0, B - 1):
        Jo, trajo, _ = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, model), orc.MPCState.from_array(states[b]), U[b].astype(np.float64))
        k = int(np.argmin(Jo))
        if dtype == "f64":
            assert int(res[b, 1]) == k
            np.testing.assert_allclose(res[b, 0], Jo[k], rtol=1e-9)
            np.testing.assert_allclose(res[b, 5:].reshape(-1, 2), trajo[k], rtol=1e-9, atol=1e-12)
            assert np.array_equal(res[b, 2:5], U[b, k, 0])
        else:
            assert int(res[b, 1]) == k or abs(Jo[int(res[b, 1])] - Jo[k]) / Jo[k] < 1e-4


def test_batched_launch_with_other_models(rv, orc):
    """The batched grid through the hiprtc and interpreter variants of the kernel."""
    import torch
    dev = torch.device("cuda", 0)
    B, K, N = 4, 96, 9
    for kw in ({"no_builtin": True}, {"force_interpreter": True}):
        cfg = rv.MPCConfig(N=N, K=K, **kw)
        states = np.empty((B, 16)); U = np.empty((B, K, N, 3))
        for b in range(B):
            states[b], U[b] = rv.synthetic_problem(K, N, seed=70 + b)
        d_states = torch.tensor(states, device=dev); d_U = torch.tensor(U, device=dev)
        with rv.Engine(cfg) as e:
            d_res = torch.empty((B, e.result_len), dtype=torch.float64, device=dev)
            e.step_batch_device(B, d_states.data_ptr(), d_U.data_ptr(), d_res.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            res = d_res.cpu().numpy()
            for b in range(B):
                one = e.step(states[b], U[b])
                assert one.cost == res[b, 0] and one.index == int(res[b, 1])
                np.testing.assert_array_equal(one.traj, res[b, 5:].reshape(-1, 2))


def test_config4_size_eight_shards_of_4096(rv, orc):
    """BASELINE configs[3] as far as one GPU can take it: K = 32768 split into 8 contiguous shards of 4096, each rolled
    out by the sharded step into its slot row, min over the 8 slot images (what the all-reduce leaves), select kernel
    == the un-sharded Engine(K=32768).step == the oracle's arg-min."""
    import torch
    G, Kl, N = 8, 4096, 20
    K = G * Kl
    state, U = rv.synthetic_problem(K, N, seed=33)
    dev = torch.device("cuda", 0)
    d_state = torch.tensor(state, device=dev); d_U = torch.tensor(U, device=dev)
    with rv.Engine(rv.MPCConfig(N=N, K=K)) as e:
        full = e.step(state, U)
        Jfull = e.rollout_costs(state, U)
    with rv.Engine(rv.MPCConfig(N=N, K=Kl)) as e:
        R = e.result_len
        slots = torch.empty((G, G, R), dtype=torch.int64, device=dev)
        for g in range(G):
            e.step_device_sharded(d_state.data_ptr(), d_U[g * Kl:(g + 1) * Kl].data_ptr(), g * Kl, g, G, slots[g].data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        reduced = slots.min(dim=0).values.contiguous()
        out = torch.empty(R, dtype=torch.float64, device=dev)
        e.select_device(reduced.data_ptr(), G, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert o[0] == full.cost and int(o[1]) == full.index and np.array_equal(o[2:5], full.u)
    np.testing.assert_array_equal(o[5:].reshape(-1, 2), full.traj)
    cfg = rv.MPCConfig(N=N, K=K)
    Jo, trajo, _ = orc.rollout_vec(oracle_cfg(orc, cfg), oracle_model(orc, rv.default_model()), orc.MPCState.from_array(state), U)
    assert int(np.argmin(Jo)) == full.index
    np.testing.assert_allclose(Jfull, Jo, rtol=1e-9)
    np.testing.assert_allclose(full.traj, trajo[full.index], rtol=1e-9, atol=1e-12)


def _reproduce_recorded_steps(rv, e, rows, pools_host, rep, feedback, steps):
    """Recorded steps of a closed loop, re-run as stand-alone launches on the state the plant rule gives them."""
    from rovmpc.closed_loop import state_of_step
    for i in steps:
        r = e.step(state_of_step(rows, rep, i, feedback), pools_host[i % len(pools_host)])
        assert r.cost == rep.cost[i] and r.index == rep.index[i] and np.array_equal(r.u, rep.u[i]), i
        np.testing.assert_array_equal(r.traj[0], rep.theta_gamma[i])
        np.testing.assert_array_equal(r.traj[1], rep.theta_gamma[i + 1])


def test_config5_size_closed_loop_10000_steps(rv):
    """BASELINE configs[4] on one GPU: 10 000 closed-loop steps of Rov_traj_gen case 12 at N=20, K=4096 with the model's own
    (theta, gamma) fed back.  The first, the last and three random recorded steps are reproduced bit for bit by stand-alone
    launches on the recorded state.  With the reference's chosen rows the fed-back loop DIVERGES -- dgamma/dt = x15 - x17 is a
    delay recurrence with a root at one, nothing pulls gamma back -- and that is asserted, not hidden: finite but astronomic
    costs in fp64 (parity is the bar, not plausibility), +inf costs (the kernel's NaN -> +inf rule) rather than garbage in
    fp32.  The same loop on a row pair with a restoring term (rows 13 / 23) keeps its cost of order one over all 10 000 steps."""
    import torch
    from rovmpc.closed_loop import run_closed_loop, closed_loop_inputs, closed_loop_pools
    rng = np.random.default_rng(5)
    spots = [0, 9999] + sorted(int(v) for v in rng.integers(1, 9999, 3))
    with rv.Engine(rv.MPCConfig(N=20, K=4096)) as e:
        rows, _ = closed_loop_inputs(e, 12, 10000)
        pools = closed_loop_pools(e, 8, 0)
        run_closed_loop(e, 12, 100, feedback=True, pools=pools)
        for mode in ("per_step", "pipelined"):
            rep = run_closed_loop(e, 12, 10000, feedback=True, mode=mode, pools=pools)
            assert rep.steps == 10000 and rep.u.shape == (10000, 3) and rep.theta_gamma.shape == (10001, 2)
            assert np.isfinite(rep.cost).all() and np.isfinite(rep.theta_gamma).all()
            assert rep.wall_s < 2.0 and rep.real_time_factor > 80                 # ~0.2 s of GPU time (13-20 us per step)
            _reproduce_recorded_steps(rv, e, rows, pools.cpu().numpy(), rep, True, spots)
            # the divergence of the chosen rows under feedback, stated: gamma drifts away and the cost with it
            assert abs(rep.theta_gamma[-1, 1]) > 1e3 and rep.cost[-1] > 1e6
            assert np.abs(rep.theta_gamma[:300]).max() < 10.0                     # the first few hundred steps are tame
    with rv.Engine(rv.MPCConfig(N=20, K=4096, dtype="f32")) as e:
        rep32 = run_closed_loop(e, 12, 10000, feedback=True)
        assert not np.isnan(rep32.cost).any()                                     # never NaN: NaN costs are +inf by the kernel's rule
        assert np.isinf(rep32.cost[-1]) or rep32.cost[-1] > 1e6                   # overflowed (or astronomic), visibly
        assert np.isfinite(rep32.cost[:300]).all()
        assert ((rep32.index >= 0) & (rep32.index < 4096)).all()
    with rv.Engine(rv.MPCConfig(N=20, K=4096), rv.default_model(13, 23)) as e:    # dgamma/dt row with tanh(x17) pulling back
        assert e.model_path == "jit"
        rows, _ = closed_loop_inputs(e, 12, 10000)
        pools = closed_loop_pools(e, 8, 0)
        rep = run_closed_loop(e, 12, 10000, feedback=True, pools=pools)
        assert np.isfinite(rep.cost).all() and rep.cost.max() < 50.0, rep.cost.max()
        _reproduce_recorded_steps(rv, e, rows, pools.cpu().numpy(), rep, True, spots)


@pytest.mark.parametrize("feedback,steps", [(True, 300), (False, 60)])
def test_closed_loop_equals_the_oracle(rv, orc, feedback, steps):
    """The closed loop -- plant rule + first-node feedback + per-step arg-min -- against the oracle's restatement
    (oracle.closed_loop: rollout_vec per step), Rov_traj_gen case 12 at K = 512, in every device form: launch per step,
    pipelined with the state handed over on the GPU, and the sharded step on a one-rank communicator.  Same chosen candidate
    and control at every step, costs and (theta, gamma) to 1e-9."""
    from rovmpc.closed_loop import run_closed_loop, closed_loop_inputs, closed_loop_pools
    from rovmpc.sharded import NativeShardedMPC
    cfg = rv.MPCConfig(N=20, K=512)
    with rv.Engine(cfg) as e:
        rows, _ = closed_loop_inputs(e, 12, steps)
        pools = closed_loop_pools(e, 8, 0)
        want = orc.closed_loop(oracle_cfg(orc, cfg), oracle_model(orc, rv.default_model()), rows, pools.cpu().numpy(), feedback)
        reps = {m: run_closed_loop(e, 12, steps, feedback=feedback, mode=m, pools=pools) for m in ("per_step", "pipelined")}
        smpc = NativeShardedMPC(e, rank=0, world=1)
        reps["sharded"] = run_closed_loop(e, 12, steps, feedback=feedback, pools=pools)
        smpc.close()
    for name, rep in reps.items():
        assert np.array_equal(rep.index, want["index"]), name
        assert np.array_equal(rep.u, want["u"]), name
        np.testing.assert_allclose(rep.cost, want["cost"], rtol=1e-9, err_msg=name)
        np.testing.assert_allclose(rep.theta_gamma, want["theta_gamma"], rtol=1e-9, atol=1e-12, err_msg=name)


@pytest.mark.parametrize("feedback", [True, False])
def test_sharded_closed_loop_equals_the_unsharded_loop(rv, feedback):
    """BASELINE configs[4] as it is asked (the sharded step inside the loop), on a one-rank RCCL communicator:
    rovmpc_closed_loop_device's communicator branch -- plant update, rollout into the slot row, ncclAllReduce, select, join per
    step -- must give the un-sharded loop's records bit for bit over 250 steps, also for a shard that does not start at
    candidate 0 (global indices = local + k_offset).  A hand-off that times out inside the loop is an error of the call,
    never a silently wrong record, and the handle recovers."""
    from rovmpc.closed_loop import run_closed_loop, closed_loop_pools
    from rovmpc.sharded import NativeShardedMPC
    T = 250
    with rv.Engine(rv.MPCConfig(N=20, K=1024)) as e:
        pools = closed_loop_pools(e, 8, 3)
        plain = run_closed_loop(e, 12, T, feedback=feedback, pools=pools)
        plain20 = run_closed_loop(e, 12, 20, feedback=feedback, pools=pools)         # (the trajectory table depends on the loop's length)
        smpc = NativeShardedMPC(e, rank=0, world=1)
        assert e.has_comm
        for k_offset in (0, 5 * 1024):
            rep = run_closed_loop(e, 12, T, feedback=feedback, pools=pools, k_offset=k_offset)
            assert np.array_equal(rep.cost, plain.cost) and np.array_equal(rep.u, plain.u)
            assert np.array_equal(rep.theta_gamma, plain.theta_gamma)
            assert np.array_equal(rep.index, plain.index + k_offset)
        # one rollout of the loop never publishes its row: the collective of that step gives up after the hand-off time-out
        e.set_option("handoff_timeout_ms", 150.0)
        e.set_option("inject_skip_rolled", 1)
        # (with feedback the loop joins every step and stops at the collective's give-up; without, the rollouts queued behind
        # it give up on their slot rows first -- either way the call fails with the hand-off's reason)
        with pytest.raises(rv.RovmpcError, match="GPU-side hand-off gave up"):
            run_closed_loop(e, 12, 40, feedback=feedback, pools=pools)
        e.set_option("handoff_timeout_ms", 10000.0)
        again = run_closed_loop(e, 12, T, feedback=feedback, pools=pools)
        assert np.array_equal(again.cost, plain.cost) and np.array_equal(again.index, plain.index)
        smpc.close()
        assert not e.has_comm
        after = run_closed_loop(e, 12, 20, feedback=feedback, pools=pools)           # the handle is a plain one again
        assert np.array_equal(after.cost, plain20.cost) and np.array_equal(after.index, plain20.index)


@pytest.mark.parametrize("feedback", [False, True])
@pytest.mark.parametrize("K,N,steps,kw", [(4096, 20, 300, {}), (1024, 20, 64, {"dtype": "f32"}), (512, 12, 40, {"no_builtin": True}),
                                           (4096, 20, 40, {"vt_mode": 0})])
def test_pipelined_closed_loop_equals_launch_per_step(rv, K, N, steps, kw, feedback):
    """rovmpc_closed_loop_pipelined_device (one launch per step on two alternating streams, the state handed over on the GPU
    between launches) must reproduce the records of the launch-per-step loop bit for bit, with measured rows and with the
    model's own (theta, gamma) fed back."""
    from rovmpc.closed_loop import run_closed_loop
    mode = "pipelined"
    with rv.Engine(rv.MPCConfig(N=N, K=K, **kw)) as e:
        e.set_option("handoff_timeout_ms", 2000.0)
        a = run_closed_loop(e, 12, steps, feedback=feedback, mode="per_step")
        b = run_closed_loop(e, 12, steps, feedback=feedback, mode=mode)
        c = run_closed_loop(e, 12, steps, feedback=feedback, mode=mode)              # epochs advance, buffers are reusable
        e.device_status()                                                             # no hand-off gave up
    for r in (b, c):
        if kw.get("dtype") == "f32":
            # two instantiations of the fp32 body: the compiler's fma contraction may differ in the last bit
            # (with feedback a last-bit difference is carried from step to step, hence the looser bar)
            np.testing.assert_allclose(r.cost, a.cost, rtol=1e-4 if feedback else 2e-6)
            np.testing.assert_allclose(r.theta_gamma, a.theta_gamma, rtol=1e-4 if feedback else 2e-6, atol=1e-7)
        else:
            assert np.array_equal(a.cost, r.cost) and np.array_equal(a.u, r.u) and np.array_equal(a.theta_gamma, r.theta_gamma)
    assert np.isfinite(a.cost).all()
    assert np.array_equal(b.cost, c.cost) and np.array_equal(b.u, c.u)


def test_batched_replay_of_measured_rows_equals_the_step_by_step_loop(rv):
    """Measured rows make the steps independent: n_pools of them per batched launch give the records of the per-step loop."""
    from rovmpc.closed_loop import run_closed_loop
    with rv.Engine(rv.MPCConfig(N=20, K=4096)) as e:
        a = run_closed_loop(e, 12, 100, feedback=False, mode="per_step")
        b = run_closed_loop(e, 12, 100, feedback=False, mode="batched")            # 12 launches of 8 + 4 singles
        assert np.array_equal(a.cost, b.cost) and np.array_equal(a.u, b.u) and np.array_equal(a.theta_gamma, b.theta_gamma)
        with pytest.raises(ValueError):
            run_closed_loop(e, 12, 10, feedback=True, mode="batched")


def test_pipelined_closed_loop_refuses_what_it_cannot_hold(rv):
    from rovmpc.closed_loop import run_closed_loop
    with rv.Engine(rv.MPCConfig(N=20, K=16384)) as e:            # two grids of 1024 workgroups do not fit the chip at once
        with pytest.raises(rv.RovmpcError, match="resident"):
            run_closed_loop(e, 12, 10, mode="pipelined")
    with rv.Engine(rv.MPCConfig(N=8, K=64, force_interpreter=True)) as e:
        with pytest.raises(rv.RovmpcError, match="interpreter"):
            run_closed_loop(e, 12, 10, mode="pipelined")
    assert "rovmpc_closed_loop_persistent_device" not in rv.exported_symbols()     # removed in round 3 (slower than this form)


def test_bench_two_ranks_on_one_gpu_over_gloo(rv):
    """The N > 1 step with real kernels and two real processes: `bench.py --gpus 2 --backend gloo --devices 0,0` starts its
    own two ranks, both on GPU 0 (RCCL refuses a shared GPU, so the slot image crosses the host through gloo); every rank
    must end with the same global record, and it must be the arg-min over both shards."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--devices", "0,0",
                        "--steps", "30", "--warmup", "5", "--K", "1024", "--no-kernel-timing"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_agree"] is True and d["config"]["K_global"] == 2048
    # BASELINE config 5 in its sharded form rides on the same line: the closed loop over the sharded step, every rank ending
    # with the same records
    cl = d["closed_loop"]
    assert cl["n_gpus"] == 2 and cl["ranks_agree"] is True and cl["steps"] >= 100 and cl["us_per_step"] > 0 and cl["all_costs_finite"]
    assert d["config"]["value_is_fallback"] is True and "strict_bracket" in d          # gloo: not the library's own RCCL path
    # the same 2 x 1024 candidates un-sharded on one handle: rank r's last batch is pool (steps - 1) % 8 of seed 20250523 + 1000 r + p
    pidx = (30 - 1) % 8
    state, _ = rv.synthetic_problem(1, 20)
    U = np.concatenate([rv.synthetic_problem(1024, 20, seed=20250523 + 1000 * r + pidx)[1] for r in range(2)])
    with rv.Engine(rv.MPCConfig(N=20, K=2048)) as e:
        want = e.step(state, U)
    assert d["best"]["index"] == want.index and d["best"]["cost"] == want.cost


def test_bench_checks_the_native_collective_path_and_records_a_fallback(rv):
    """A multi-rank bench run checks the library's own RCCL path against torch.distributed's collective before timing it and
    falls back -- on every rank, with the reason in the line -- when the check fails.  Rehearsed on a one-rank communicator
    with the all-reduce kept: the check passes (native path stays), and with the failure hook it switches and says so."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = {}
    for hook in ("1", "fail", "abort"):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-collective", "--steps", "20", "--warmup", "5",
                            "--K", "1024", "--no-kernel-timing", "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=600, env={**base, "ROVMPC_BENCH_TEST_VALIDATE": hook, "MASTER_PORT": "29547"})
        assert p.returncode == 0, p.stderr[-3000:]
        out[hook] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    ok, bad = out["1"], out["fail"]
    assert ok["config"]["collective"].startswith("ncclAllReduce") and ok["config"]["collective_fallback_reason"] is None
    assert bad["config"]["collective"].startswith("torch.distributed") and "failed its check" in bad["config"]["collective_fallback_reason"]
    assert ok["best"] == bad["best"]          # either path ends on the same global record
    # a check that does not finish: the timer thread aborts the communicators (rovmpc_comm_abort), the run goes on
    aborted = out["abort"]
    assert aborted["config"]["collective"].startswith("torch.distributed") and "aborted" in aborted["config"]["collective_fallback_reason"]
    assert aborted["best"] == ok["best"]
    assert ok["config"]["hw_queues"] == "8"


def test_handoff_timeouts_are_errors_not_wrong_records(rv):
    """The sharded step's GPU-side waits (collective <- rollout row, rollout <- select that frees the slot row) give up
    after handoff_timeout_ms.  A give-up must surface as an error of the synchronising call and a NaN cost in the affected
    record, never as a silently reused row.  Injected on a one-rank RCCL communicator."""
    import torch
    from rovmpc.sharded import NativeShardedMPC
    cfg = rv.MPCConfig(N=20, K=512)
    dev = torch.device("cuda", 0)
    state, U = rv.synthetic_problem(cfg.K, cfg.N, seed=5)
    d_state = torch.tensor(state, device=dev); d_U = torch.tensor(U, device=dev)
    with rv.Engine(cfg) as e:
        want = e.step(state, U)
        smpc = NativeShardedMPC(e, rank=0, world=1)
        e.set_option("handoff_timeout_ms", 200.0)
        rec = smpc.step_device(d_state, d_U); smpc.synchronize()
        assert rec.cpu().numpy()[0] == want.cost
        # (1) one rollout does not publish its row: the collective of that step gives up -> error + NaN record
        e.set_option("inject_skip_rolled", 1)
        bad = smpc.step_device(d_state, d_U)
        with pytest.raises(rv.RovmpcError, match="never saw its rollout"):
            smpc.synchronize()
        assert np.isnan(bad.cpu().numpy()[0])
        # the error word was consumed: the next steps are clean again
        good = [smpc.step_device(d_state, d_U) for _ in range(4)]
        smpc.synchronize()
        assert all(g.cpu().numpy()[0] == want.cost for g in good)
        # (2) one select does not free its slot row: four steps later the rollout that needs the row gives up
        e.set_option("inject_skip_consumed", 1)
        outs = [smpc.step_device(d_state, d_U) for _ in range(4)]      # the step that loses its 'consumed' is itself fine
        smpc.synchronize()
        assert all(o.cpu().numpy()[0] == want.cost for o in outs)
        late = smpc.step_device(d_state, d_U)                          # same slot again: its row was never freed
        with pytest.raises(rv.RovmpcError, match="frees its slot row"):
            smpc.synchronize()
        assert np.isnan(late.cpu().numpy()[0])
        again = smpc.step_device(d_state, d_U); smpc.synchronize()
        assert again.cpu().numpy()[0] == want.cost
        smpc.close()



def test_mpc_step_surface_and_closed_loop(rv):
    mpc = rv.MPC(N=20, K=512)
    state, _ = rv.synthetic_problem(512, 20)
    u = mpc.step(state)
    assert u.shape == (3,) and mpc.last.traj.shape == (21, 2) and np.isfinite(mpc.last.cost)
    u2 = mpc.step(rv.MPCState(P1=state[3:6], V1=state[6:9], theta=-0.03, gamma=-0.05))
    assert u2.shape == (3,)
    from rovmpc.closed_loop import run_closed_loop, closed_loop_inputs
    rep = run_closed_loop(mpc.engine, exp_case=12, n_steps=50)
    assert rep.steps == 50 and rep.u.shape == (50, 3) and np.isfinite(rep.cost).all()
    assert rep.real_time_factor > 0
    # the device-side loop equals stepping by hand with the same plant rule (both modes)
    import torch
    eng = mpc.engine
    rows, _ = closed_loop_inputs(eng, 12, 50)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    mean = torch.tensor(eng.model.mean[3:6], device="cuda"); scale = torch.tensor(eng.model.scale[3:6], device="cuda")
    pools = (mean + scale * torch.randn((8, 512, 20, 3), generator=g, device="cuda", dtype=torch.float64)).cpu().numpy()
    for fb in (False, True):
        rep = run_closed_loop(eng, exp_case=12, n_steps=50, feedback=fb)
        st = rows[0].copy()
        for i in range(6):
            if fb and i > 0:
                st[0:12] = rows[i][0:12]
            else:
                st = rows[i].copy()
            r = eng.step(st, pools[i % 8])
            assert r.cost == rep.cost[i] and np.array_equal(r.u, rep.u[i])
            np.testing.assert_array_equal(r.traj[1], rep.theta_gamma[i + 1])
            if fb:
                th, ga = st[12], st[13]
                st = st.copy(); st[14], st[15] = th, ga; st[12], st[13] = r.traj[1]
    mpc.close()


def test_closed_loop_second_order_model(rv):
    """Device-side closed loop with the second-order generation (measured rows carry the rates): equals stepping by
    hand; model feedback is refused (it would need the rates of the winner)."""
    import torch
    from rovmpc.closed_loop import run_closed_loop, closed_loop_inputs, velocity_prior
    eng = rv.Engine(rv.MPCConfig(N=10, K=256, feature_map=rv.FEATURES_GEN3), rv.generation3_model())
    rep = run_closed_loop(eng, exp_case=12, n_steps=12)
    assert np.isfinite(rep.cost).all()
    rows, _ = closed_loop_inputs(eng, 12, 12)
    vm, vs = velocity_prior(eng)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    pools = (torch.tensor(vm, device="cuda") + torch.tensor(vs, device="cuda")
             * torch.randn((8, 256, 10, 3), generator=g, device="cuda", dtype=torch.float64)).cpu().numpy()
    for i in range(4):
        r = eng.step(rows[i], pools[i % 8])
        assert r.cost == rep.cost[i] and np.array_equal(r.u, rep.u[i])
    with pytest.raises(rv.RovmpcError):
        run_closed_loop(eng, exp_case=12, n_steps=4, feedback=True)
    eng.close()


def test_lagrangian_residuals_and_rollout(rv, orc, golden_dir):
    """SURVEY 8f N4, Lagrangian half: Euler-Lagrange residuals on the GPU against the residual files the reference's own
    runs stored (outputs/Lg_C6_*), and against sympy's route on synthetic Lagrangians; the forward integration of
    evaluate_lagrangian_on_test.py:59-68 against its restatement, 5 initial states in one launch."""
    g = np.load(os.path.join(golden_dir, "kat_lagrangian.npz"))
    series = [g[k] for k in ("theta", "gamma", "dtheta", "dgamma", "ddtheta", "ddgamma")]
    for tag in ("full", "split_hy", "split"):
        r_th, r_ga = rv.el_residuals(str(g[f"expr_{tag}"]), *series)
        np.testing.assert_allclose(r_th, g[f"residual_theta_{tag}"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(r_ga, g[f"residual_gamma_{tag}"], rtol=1e-12, atol=1e-13)
    rows = g["synth_rows"]
    for j, txt in enumerate(g["synth_exprs"]):
        r_th, r_ga = rv.el_residuals(str(txt), *rows.T)
        np.testing.assert_allclose(r_th, g[f"synth_res_theta_{j}"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r_ga, g[f"synth_res_gamma_{j}"], rtol=1e-10, atol=1e-12)
    y0 = g["synth_y0"]
    for j in range(2):
        th, ga, vt, vg = rv.lagrangian_rollout(str(g["synth_exprs"][j]), g["synth_time"], y0[:, 0], y0[:, 1], y0[:, 2], y0[:, 3])
        want = g[f"synth_rollout_{j}"]                                        # (5, 4, T)
        # (one synthetic system blows up near the end of the grid; there rounding differences are amplified without bound,
        # so rows are compared up to the point where the reference restatement itself leaves |x| < 10)
        tame = np.all(np.abs(want) < 10.0, axis=1)                              # (5, T)
        tame = np.logical_and.accumulate(tame, axis=1)
        assert tame[:, :40].all()
        for got, k in ((th, 0), (ga, 1), (vt, 2), (vg, 3)):
            np.testing.assert_allclose(got[tame], want[:, k, :][tame], rtol=1e-7, atol=1e-10)
            np.testing.assert_allclose(got[:, :40], want[:, k, :40], rtol=1e-10, atol=1e-13)
        one = rv.lagrangian_rollout(str(g["synth_exprs"][j]), g["synth_time"], *y0[3])
        np.testing.assert_allclose(one[0][tame[3]], want[3, 0][tame[3]], rtol=1e-7, atol=1e-10)
    # the reference's own best Lagrangian (dtheta^2 + dgamma^2): zero accelerations, straight lines
    th, ga, vt, vg = rv.lagrangian_rollout(str(g["expr_full"]), g["time"], g["theta"][0], g["gamma"][0], g["dtheta"][0], g["dgamma"][0])
    a_th, a_ga = orc.lagrangian_accelerations(str(g["expr_full"]))
    want = orc.lagrangian_rollout(a_th, a_ga, g["time"], g["theta"][0], g["gamma"][0], g["dtheta"][0], g["dgamma"][0])
    np.testing.assert_allclose(th, want[0], rtol=1e-12, atol=1e-15); np.testing.assert_allclose(vg, want[3], rtol=1e-12)
    with pytest.raises(rv.ExpressionError, match="cannot be solved"):
        rv.lagrangian_rollout("0.5*x2**2 + 0.5*x3**2 + 0.3*x2*x3*cos(x0)", g["synth_time"], 0.1, 0.1, 0.0, 0.0)


@pytest.mark.parametrize("fused", [True, False])
def test_mpc_step_with_device_side_sampling(rv, orc, fused):
    """MPC(device_sampling=True): one library call per step draws the candidates on the GPU (Philox4x32-10 + Box-Muller),
    rolls them out and returns the record.  The tensor is exactly the oracle's restatement of the law for (seed, step);
    the returned control is the arg-min of those candidates; warm start pins candidate 0; fp32 too.  fused: the compiled-in
    kernel draws its candidates itself (no tensor in HBM); otherwise (any other model path) sampler kernel + rollout kernel."""
    if not fused:
        _orig = rv.MPC
        class _M(_orig):                                   # same rows through hiprtc: the two-kernel path
            def __init__(self, *a, **kw):
                kw["no_builtin"] = True
                super().__init__(*a, **kw)
        MPC = _M
    else:
        MPC = rv.MPC
    mpc = MPC(N=12, K=256, device_sampling=True, sampler=rv.DeviceGaussianSampler(seed=99))
    assert mpc.engine.model_path == ("builtin" if fused else "jit")
    state, _ = rv.synthetic_problem(256, 12)
    model = rv.default_model()
    u = mpc.step(state)
    U = mpc.engine.sampled_candidates()
    Uo = orc.sample_candidates(99, 0, 256, 12, model.mean[3:6], model.scale[3:6])
    np.testing.assert_allclose(U, Uo, rtol=1e-12, atol=1e-9)
    Jo, trajo, _ = orc.rollout_vec(oracle_cfg(orc, mpc.cfg), oracle_model(orc, model), orc.MPCState.from_array(state), U)
    k = int(np.argmin(Jo))
    assert mpc.last.index == k and np.array_equal(u, U[k, 0])
    assert mpc.last.cost == pytest.approx(Jo[k], rel=RTOL)
    np.testing.assert_allclose(mpc.last.traj, trajo[k], rtol=RTOL, atol=1e-13)
    best = U[k].copy()
    u2 = mpc.step(state)
    U2 = mpc.engine.sampled_candidates()
    np.testing.assert_array_equal(U2[0], np.vstack([best[1:], best[-1:]]))       # shifted previous optimum
    np.testing.assert_allclose(U2[1:], orc.sample_candidates(99, 1, 256, 12, model.mean[3:6], model.scale[3:6])[1:], rtol=1e-12, atol=1e-9)
    Jo2, _, _ = orc.rollout_vec(oracle_cfg(orc, mpc.cfg), oracle_model(orc, model), orc.MPCState.from_array(state), U2)
    assert mpc.last.index == int(np.argmin(Jo2)) and np.array_equal(u2, U2[mpc.last.index, 0])
    # the same (seed, step) gives the same step on a fresh controller; another seed does not
    mpc_b = MPC(N=12, K=256, device_sampling=True, sampler=rv.DeviceGaussianSampler(seed=99))
    assert np.array_equal(mpc_b.step(state), u)
    mpc_c = MPC(N=12, K=256, device_sampling=True, sampler=rv.DeviceGaussianSampler(seed=100))
    mpc_c.step(state)
    assert not np.array_equal(mpc_c.engine.sampled_candidates(), U)
    # fp32 tensor: the fp64 draw rounded once
    mpc_f = MPC(N=12, K=256, dtype="f32", device_sampling=True, sampler=rv.DeviceGaussianSampler(seed=99))
    mpc_f.step(state)
    np.testing.assert_allclose(mpc_f.engine.sampled_candidates(), Uo.astype(np.float32), rtol=2e-7, atol=1e-5)
    # the stand-alone sampler entry fills a caller's device tensor with the same law
    import torch
    dU = torch.empty((256, 12, 3), dtype=torch.float64, device="cuda:0")
    mpc.engine.sample_candidates_device(99, 0, model.mean[3:6], model.scale[3:6], dU.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dU.cpu().numpy(), U)
    # the C2 size through the same call (its latency is a reported number of tools/mpc_step_rate.py, not a test bar)
    mpc_big = MPC(N=20, K=4096, device_sampling=True)
    st, _ = rv.synthetic_problem(1, 20)
    for _ in range(5):
        ub = mpc_big.step(st)
    assert ub.shape == (3,) and np.isfinite(mpc_big.last.cost)
    for m in (mpc, mpc_b, mpc_c, mpc_f, mpc_big):
        m.close()


def test_device_sampling_warm_start_survives_leaving_the_fused_path(rv, orc):
    """A fused step (compiled-in model: the kernel draws its own candidates) keeps the winner's sequence on the device only.
    When the handle leaves the fused path -- here by rovmpc_timing_enable -- the next step's candidate 0 must still be that
    winner shifted by one step (round-2 advisory: it read a never-written tensor), and the steps after it chain normally."""
    mpc = rv.MPC(N=12, K=256, device_sampling=True, sampler=rv.DeviceGaussianSampler(seed=7))
    model = rv.default_model()
    state, _ = rv.synthetic_problem(256, 12)
    mpc.step(state); mpc.step(state)                            # two fused steps (both parities of the winner buffer)
    U = mpc.engine.sampled_candidates()
    best = U[mpc.last.index].copy()
    mpc.engine.timing_enable(8)                                 # two-kernel path from here on
    for step in (2, 3):
        u = mpc.step(state)
        Un = mpc.engine.sampled_candidates()
        np.testing.assert_array_equal(Un[0], np.vstack([best[1:], best[-1:]]))
        np.testing.assert_allclose(Un[1:], orc.sample_candidates(7, step, 256, 12, model.mean[3:6], model.scale[3:6])[1:], rtol=1e-12, atol=1e-9)
        Jo, _, _ = orc.rollout_vec(oracle_cfg(orc, mpc.cfg), oracle_model(orc, model), orc.MPCState.from_array(state), Un)
        assert mpc.last.index == int(np.argmin(Jo)) and np.array_equal(u, Un[mpc.last.index, 0])
        best = Un[mpc.last.index].copy()
    mpc.close()


def test_error_behaviour(rv):
    with pytest.raises(rv.RovmpcError):
        rv.Engine(rv.MPCConfig(N=0, K=4))
    with pytest.raises(rv.RovmpcError):
        rv.Engine(rv.MPCConfig(N=4, K=4, n_shape_pts=1))
    with rv.Engine(rv.MPCConfig(N=4, K=8)) as e:
        state, U = rv.synthetic_problem(8, 4)
        with pytest.raises(ValueError):
            e.step(state, U[:4])
        with pytest.raises(ValueError):
            e.step(state[:5], U)
    with rv.Engine(rv.MPCConfig(N=4, K=8, vt_mode=rv.VT_TABLE)) as e:
        with pytest.raises(rv.RovmpcError):
            e.step(state, U)                         # rotation table not set
    with pytest.raises(rv.ExpressionError):
        rv.DynamicsModel(np.zeros(18), np.ones(18), "x3 + foo(x1)", "x15")
    with pytest.raises(rv.ExpressionError):
        rv.DynamicsModel(np.zeros(18), np.ones(18), "x3 + x99", "x15")


def test_plain_c_consumer_of_the_abi(rv, tmp_path):
    """A C program (gcc, no Python, no torch) links librovmpc.so and drives the hot path."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(rv.LIB_PATH)
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.run(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "c_abi_smoke.c"),
                    "-o", exe, "-L", libdir, "-lrovmpc", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "c_abi_smoke ok" in r.stdout
