#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference, sympy, pandas).  Nothing here
travels to the GPU box except the emitted data files.  What is executed from the
reference:

* ``import main_fun`` (works: the reference's ``wandb/`` run directory resolves as a
  namespace package) -> rodrigues_rotation, transform_catenary, solve_catenary,
  build_theta_features (tension rule), extract_features, integrate_theta_gamma;
* the *function definitions* ``rk4_integration`` (simulate_rk4_theta_gamma.py:52-68)
  and ``extract_features`` (simply.py:15-41), pulled out of their scripts with ``ast``
  and exec'd here, because the scripts themselves cannot be imported (module-level
  ``pd.read_csv`` of data files absent from the snapshot);
* the ``sympy_format`` strings of saved_models/equations_*.csv, lambdified with sympy.

What is NOT executed: ``saved_models/*.pkl`` are pickles shipped inside the reference
and are never unpickled.  The StandardScaler's ``mean_``/``var_``/``scale_`` float64
arrays are recovered from ``scaler.pkl`` by a raw byte scan (joblib stores ndarray
payloads verbatim), anchored on the attribute-name markers and cross-checked
(scale**2 == var, SURVEY.md section 8 A1 values).

``pympc.models.catenary.Catenary`` is absent from the reference; transform_catenary
KATs are generated with the oracle's own Catenary passed in as ``catenary_fn`` -- that
pins everything in transform_catenary except the absent third-party callable.
"""
import ast
import csv
import json
import os
import shutil
import struct
import sys
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")


def scaler_arrays_from_raw_bytes(path, n_feat):
    """(mean, var, scale, offsets) of a joblib-dumped StandardScaler, read as raw float64 payloads."""
    b = open(path, "rb").read()

    def arr_after(marker):
        i = b.find(marker)
        assert i >= 0, marker
        j = b.find(b"numpy_array_alignment_bytes", i)
        if j < 0 or j - i > 400:
            j = i
        for start in range(j, min(j + 200, len(b) - 8 * n_feat)):
            vals = struct.unpack("<%dd" % n_feat, b[start:start + 8 * n_feat])
            if all(np.isfinite(v) and 1e-12 < abs(v) < 1e12 for v in vals):
                return start, np.array(vals)
        raise AssertionError(marker)

    o_mean, mean = arr_after(b"mean_")
    o_var, var = arr_after(b"var_")
    o_scale, scale = arr_after(b"scale_")
    assert np.allclose(scale ** 2, var, rtol=1e-12), "scale^2 != var: wrong offsets"
    return mean, var, scale, [o_mean, o_var, o_scale]


def scaler_from_raw_bytes(path):
    b = open(path, "rb").read()
    n_feat = 18

    def arr_after(marker):
        i = b.find(marker)
        assert i >= 0, marker
        # the payload is the first 8-byte-aligned-looking block of n_feat finite doubles
        # after the NumpyArrayWrapper header that follows the marker
        j = b.find(b"numpy_array_alignment_bytes", i)
        if j < 0 or j - i > 400:
            j = i
        best = None
        for start in range(j, min(j + 200, len(b) - 8 * n_feat)):
            vals = struct.unpack("<%dd" % n_feat, b[start:start + 8 * n_feat])
            if all(np.isfinite(v) and 1e-12 < abs(v) < 1e12 for v in vals):
                best = (start, np.array(vals))
                break
        assert best is not None, marker
        return best

    o_mean, mean = arr_after(b"mean_")
    o_var, var = arr_after(b"var_")
    o_scale, scale = arr_after(b"scale_")
    assert np.allclose(scale ** 2, var, rtol=1e-12), "scale^2 != var: wrong offsets"
    # SURVEY section 8 A1 anchors
    assert mean[3] == 80.85390753943355 and scale[3] == 108.48977412955092
    assert mean[15] == -0.052162842559795954 and scale[15] == 0.01732537745796056
    assert mean[16] == -0.03419530357326386 and scale[16] == 0.028848274510329956
    assert mean[17] == -0.052169946475892334 and scale[17] == 0.017328840568644522
    return {"n_features": n_feat, "mean": mean.tolist(), "var": var.tolist(),
            "scale": scale.tolist(), "byte_offsets": [o_mean, o_var, o_scale],
            "n_samples_seen": 861,
            "source": "saved_models/scaler.pkl raw float64 payloads (not unpickled)"}


def read_equations():
    out = {}
    for which in ("dtheta_dt", "dgamma_dt"):
        rows = []
        with open(f"{REF}/saved_models/equations_{which}.csv") as f:
            for r in csv.DictReader(f):
                rows.append({"complexity": int(r["complexity"]), "loss": float(r["loss"]),
                             "score": float(r["score"]), "equation": r["equation"],
                             "sympy_format": r["sympy_format"]})
        txt = open(f"{REF}/saved_models/eq_{which}.txt").read()
        chosen = int(txt.split("\n")[0].split()[-1])      # "complexity   13"
        out[which] = {"chosen_complexity": chosen, "rows": rows}
    return out


def lambdify_rows(rows, n_feat=18):
    import sympy
    syms = sympy.symbols(" ".join(f"x{i}" for i in range(n_feat)))
    fns = []
    for r in rows:
        e = sympy.sympify(r["sympy_format"])
        fns.append(sympy.lambdify(syms, e, "numpy"))
    return fns


def extract_function(path, name):
    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return ast.get_source_segment(src, node)
    raise KeyError(name)


class LambdaModel:
    def __init__(self, fn):
        self.fn = fn

    def predict(self, X):
        X = np.asarray(X, float)
        out = self.fn(*[X[:, i] for i in range(X.shape[1])])
        return np.broadcast_to(np.asarray(out, float), (X.shape[0],)).copy()


def main():
    import matplotlib
    matplotlib.use("Agg")
    import pandas as pd
    import main_fun
    from oracle import rovmpc_oracle as orc

    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(0)

    # 1. scaler
    scaler = scaler_from_raw_bytes(f"{REF}/saved_models/scaler.pkl")
    json.dump(scaler, open(f"{OUT}/scaler.json", "w"), indent=1)
    mean = np.array(scaler["mean"]); scale = np.array(scaler["scale"])

    # 2. equations
    eqs = read_equations()
    json.dump(eqs, open(f"{OUT}/equations.json", "w"), indent=1)

    # 3. dynamics KATs: every row of both Pareto fronts, 256 scaled feature rows
    Xraw = mean + scale * rng.standard_normal((256, 18))
    Xs = (Xraw - mean) / scale
    fth = lambdify_rows(eqs["dtheta_dt"]["rows"]); fga = lambdify_rows(eqs["dgamma_dt"]["rows"])
    cols = [Xs[:, i] for i in range(18)]
    out_th = np.stack([np.broadcast_to(np.asarray(f(*cols), float), (256,)) for f in fth])
    out_ga = np.stack([np.broadcast_to(np.asarray(f(*cols), float), (256,)) for f in fga])
    np.savez(f"{OUT}/kat_dynamics.npz", Xraw=Xraw, Xs=Xs, out_theta=out_th, out_gamma=out_ga)

    # 4. solve_catenary + tension rule (main_fun.build_theta_features column 5)
    l = np.concatenate([[1.0, 1.41421356, 2.0, 2.5, 2.9, 0.5, 3.5, 0.6, 0.7, 2.99, 3.0, 1e-3],
                        rng.uniform(0.05, 3.2, 116)])
    dH = np.concatenate([[0, -1, 0.5, -0.3, 0.1, 0, 0, 0, 2.5, 0.0, 0.0, 1.0],
                         rng.uniform(-2.0, 2.0, 116)])
    L = 3.0
    C = main_fun.solve_catenary(l, dH, L)
    n = len(l)
    df = pd.DataFrame({
        "rod_end X": 0.0, "rod_end Y": 0.0, "rod_end Z": 0.0,
        "robot_cable_attach_point X": l * 1000.0, "robot_cable_attach_point Y": 0.0,
        "robot_cable_attach_point Z": dH * 1000.0,
        "rob_cor_speed X": rng.standard_normal(n), "rob_cor_speed Y": rng.standard_normal(n),
        "rob_cor_speed Z": rng.standard_normal(n),
        "Theta": rng.standard_normal(n) * 0.1, "Gamma": rng.standard_normal(n) * 0.1,
        "Time": np.arange(n) * 0.05})
    feats = main_fun.build_theta_features(df, L, 1.521)
    np.savez(f"{OUT}/kat_solve_catenary.npz", l=feats[:, 3], dH=feats[:, 4], L=L,
             w_wet=1.521, C=main_fun.solve_catenary(feats[:, 3], feats[:, 4], L), T=feats[:, 5],
             l_in=l, dH_in=dH, C_in=C)

    # 5. rodrigues
    v = rng.standard_normal((64, 3)); ax = rng.standard_normal((64, 3)) * rng.uniform(0.1, 5, (64, 1))
    ang = rng.uniform(-np.pi, np.pi, 64)
    v[0] = [1, 1, -1]; ax[0] = np.array([1, -1, 0]) / np.sqrt(2); ang[0] = np.radians(-10)
    out = np.stack([main_fun.rodrigues_rotation(v[i], ax[i], ang[i]) for i in range(64)])
    np.savez(f"{OUT}/kat_rodrigues.npz", v=v, axis=ax, angle=ang, out=out)

    # 6. transform_catenary with the oracle Catenary as catenary_fn (M = 16)
    cat = orc.Catenary(3.0, "ENU", n_points=16)
    cases = []
    A = np.zeros(3)
    pts = [([1., 1., -1.], -10, 20), ([1., 1., 0.], 5, -15), ([0.2435, -0.7583, 0.2980], -2, -3),
           ([0., 0., -1.5], 10, 10),           # vertical: degenerate xy projection
           ([2.5, 1.5, 0.8], 12, -7),          # nearly taut
           ([3.0, 1.0, 0.0], 3, 3),            # taut -> None -> straight segment
           ([0.3, 0.1, 0.0], 4, -9)]           # root above bracket -> None
    for _ in range(25):
        p = rng.uniform(-1.5, 1.5, 3)
        pts.append((p.tolist(), float(rng.uniform(-20, 20)), float(rng.uniform(-30, 30))))
    Aarr, Barr, th, ga, o0, o1, o2, o3, npts = [], [], [], [], [], [], [], [], []
    for i, (B, t, g) in enumerate(pts):
        a = A if i % 3 else rng.uniform(-0.3, 0.3, 3)
        Bv = np.array(B, float)
        r = main_fun.transform_catenary(a.copy(), Bv.copy(), cat, np.radians(t), np.radians(g))
        Aarr.append(a); Barr.append(Bv); th.append(np.radians(t)); ga.append(np.radians(g))
        pad = lambda x: np.vstack([x, np.full((16 - len(x), 3), np.nan)])
        npts.append([len(r[0]), len(r[1])])
        o0.append(pad(r[0])); o1.append(pad(r[1])); o2.append(pad(r[2])); o3.append(pad(r[3]))
    np.savez(f"{OUT}/kat_transform_catenary.npz", A=np.array(Aarr), B=np.array(Barr),
             theta=np.array(th), gamma=np.array(ga), original=np.array(o0),
             theta_rotated=np.array(o1), theta_aligned=np.array(o2), final=np.array(o3),
             npts=np.array(npts), L=3.0, M=16)

    # 7. feature maps on a synthetic frame
    T = 64
    t = np.cumsum(rng.uniform(0.01, 0.03, T))
    fd = pd.DataFrame({
        "rod_end X": rng.normal(0, 5, T), "rod_end Y": rng.normal(0, 5, T), "rod_end Z": rng.normal(0, 5, T),
        "robot_cable_attach_point X": 243.5 + rng.normal(0, 50, T),
        "robot_cable_attach_point Y": -758.3 + rng.normal(0, 50, T),
        "robot_cable_attach_point Z": 298.0 + rng.normal(0, 50, T),
        "rob_cor_speed X": 80.85 + 108.49 * rng.standard_normal(T),
        "rob_cor_speed Y": -20.13 + 15.88 * rng.standard_normal(T),
        "rob_cor_speed Z": -18.35 + 63.13 * rng.standard_normal(T),
        "Theta": -0.0342 + 0.03 * rng.standard_normal(T),
        "Gamma": -0.0522 + 0.017 * rng.standard_normal(T), "Time": t})
    ns = {"np": np}
    exec(extract_function(f"{REF}/simply.py", "extract_features"), ns)
    X18 = ns["extract_features"](fd)
    X16 = main_fun.extract_features(fd)
    np.savez(f"{OUT}/kat_features.npz", frame=fd.values, columns=np.array(fd.columns.tolist()),
             X18=X18, X16=X16)

    # 8. rk4_integration (exec'd def) and integrate_theta_gamma on a scaled table
    ns = {"np": np}
    exec(extract_function(f"{REF}/simulate_rk4_theta_gamma.py", "rk4_integration"), ns)
    rk4 = ns["rk4_integration"]
    Xs200 = rng.standard_normal((200, 18))
    tt = np.cumsum(rng.uniform(0.005, 0.03, 200))
    i_th = [r["complexity"] for r in eqs["dtheta_dt"]["rows"]].index(eqs["dtheta_dt"]["chosen_complexity"])
    i_ga = [r["complexity"] for r in eqs["dgamma_dt"]["rows"]].index(eqs["dgamma_dt"]["chosen_complexity"])
    m_th, m_ga = LambdaModel(fth[i_th]), LambdaModel(fga[i_ga])
    m_th2, m_ga2 = LambdaModel(fth[-1]), LambdaModel(fga[-1])
    np.savez(f"{OUT}/kat_replay.npz", Xs=Xs200, time=tt, theta0=-0.0342, gamma0=-0.0522,
             rk4_theta=rk4(m_th, Xs200, tt, -0.0342), rk4_gamma=rk4(m_ga, Xs200, tt, -0.0522),
             rk4_theta_last=rk4(m_th2, Xs200, tt, -0.0342), rk4_gamma_last=rk4(m_ga2, Xs200, tt, -0.0522),
             euler=np.stack(main_fun.integrate_theta_gamma(m_th, m_ga, Xs200, tt, -0.0342, -0.0522)),
             euler_last=np.stack(main_fun.integrate_theta_gamma(m_th2, m_ga2, Xs200, tt, -0.0342, -0.0522)))

    # 8b. Kabsch rotation of cable-marker sets (velocity_transform_batch.py:8-19, exec'd def)
    ns = {"np": np}
    exec(extract_function(f"{REF}/velocity_transform_batch.py", "compute_rotation_kabsch"), ns)
    kabsch = ns["compute_rotation_kabsch"]
    Tn, Mn = 48, 16
    Pm = np.empty((Tn, Mn, 3)); Qm = np.empty((Tn, Mn, 3)); Rk = np.empty((Tn, 3, 3))
    for i in range(Tn):
        # a hanging cable: nearly planar marker set, like the mocap data the reference processes
        xs = np.linspace(0.0, rng.uniform(0.5, 2.0), Mn)
        c = rng.uniform(0.8, 3.0)
        pts = np.stack([xs, np.zeros(Mn), (np.cosh(c * (xs - xs.mean())) - 1) / c], axis=1)
        if i % 6 == 5:
            pts = rng.standard_normal((Mn, 3))                 # generic 3-D cloud
        elif i % 6 != 4:
            pts += 1e-3 * rng.standard_normal((Mn, 3))         # measurement noise (i%6==4: exactly planar)
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        Pm[i] = pts @ np.linalg.qr(rng.standard_normal((3, 3)))[0] + rng.uniform(-1, 1, 3)
        Qm[i] = (Pm[i] - Pm[i].mean(0)) @ q.T + rng.uniform(-1, 1, 3) + 1e-4 * rng.standard_normal((Mn, 3))
        Rk[i] = kabsch(Pm[i].copy(), Qm[i].copy())
    vk = rng.standard_normal((Tn, 3)) * 100
    np.savez(f"{OUT}/kat_kabsch.npz", P=Pm, Q=Qm, v=vk, R=Rk, v_out=np.einsum("tij,tj->ti", Rk, vk))

    # 9. trajectory generator outputs the reference already holds (data files)
    for c in (1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 13, 14):
        shutil.copyfile(f"{REF}/Results/Trajectory Data/rov_trajectory_exp{c}.csv",
                        f"{OUT}/rov_trajectory_exp{c}.csv")
    rng2 = np.random.default_rng(2)          # own stream: keeps sections 1-9 reproducible
    # 10. generation-2 equations (outputs/differential_training_new_feature/, 17 unscaled features)
    g2 = {}
    for which, fn in (("dtheta_dt", "dtheta_results_20250412_163500.csv"), ("dgamma_dt", "dgamma_results_20250412_163500.csv")):
        rows = []
        with open(f"{REF}/outputs/differential_training_new_feature/{fn}") as f:
            for r in csv.DictReader(f):
                rows.append({"complexity": int(r["complexity"]), "loss": float(r["loss"]), "score": float(r["score"]),
                             "equation": r["equation"], "sympy_format": r["sympy_format"]})
        txt = open(f"{REF}/outputs/differential_training_new_feature/eq_{which}_20250412_163500.txt").read()
        g2[which] = {"chosen_complexity": int(txt.split("\n")[0].split()[-1]), "rows": rows}
    json.dump(g2, open(f"{OUT}/equations_gen2.json", "w"), indent=1)
    X2 = np.hstack([rng2.normal(0, 1, (128, 3)), rng2.normal(0, 100, (128, 3)), rng2.normal(0, 300, (128, 3)),
                    rng2.normal(0, 0.6, (128, 3)), rng2.normal(0, 0.1, (128, 2)), rng2.uniform(-1, 1, (128, 3))])
    f2t = lambdify_rows(g2["dtheta_dt"]["rows"], 17); f2g = lambdify_rows(g2["dgamma_dt"]["rows"], 17)
    c2 = [X2[:, i] for i in range(17)]
    with np.errstate(all="ignore"):
        o2t = np.stack([np.broadcast_to(np.asarray(f(*c2), float), (128,)) for f in f2t])
        o2g = np.stack([np.broadcast_to(np.asarray(f(*c2), float), (128,)) for f in f2g])
    np.savez(f"{OUT}/kat_dynamics_gen2.npz", X=X2, out_theta=o2t, out_gamma=o2g)

    # 11. generation 3: second-order models on the 14 named features of features_dd (main_fun.py:811-871),
    #     outputs/dd_C6_all_50_s_20250511_013928/ (scaler.pkl read as raw bytes, 14 slots); names: dd_cluster.py:160-168 ("gama")
    rng3 = np.random.default_rng(3)
    D3 = f"{REF}/outputs/dd_C6_all_50_s_20250511_013928"
    names3 = ["theta", "gama", "dtheta", "dgamma", "v_sway", "v_surge", "a_sway", "a_surge",
              "V_x", "V_y", "V_z", "a_x", "a_y", "a_z"]
    m3, v3, s3, off3 = scaler_arrays_from_raw_bytes(f"{D3}/scaler.pkl", 14)
    json.dump({"n_features": 14, "variable_names": names3, "mean": m3.tolist(), "var": v3.tolist(), "scale": s3.tolist(),
               "byte_offsets": off3, "source": "outputs/dd_C6_all_50_s_20250511_013928/scaler.pkl raw float64 payloads (not unpickled)"},
              open(f"{OUT}/scaler_gen3.json", "w"), indent=1)
    g3 = {"variable_names": names3}
    for which, fn in (("ddtheta", "dtheta_results.csv"), ("ddgamma", "dgamma_results.csv")):
        rows = []
        with open(f"{D3}/{fn}") as f:
            for r in csv.DictReader(f):
                rows.append({"complexity": int(r["complexity"]), "loss": float(r["loss"]), "score": float(r["score"]),
                             "equation": r["equation"], "sympy_format": r["sympy_format"]})
        txt = open(f"{D3}/eq_{'dtheta_dt' if which == 'ddtheta' else 'dgamma_dt'}.txt").read()
        g3[which] = {"chosen_complexity": int(txt.split("\n")[0].split()[-1]), "rows": rows}
    json.dump(g3, open(f"{OUT}/equations_gen3.json", "w"), indent=1)
    import sympy
    syms3 = sympy.symbols(" ".join(names3))
    X3 = rng3.normal(0, 1.2, (128, 14))                      # scaled rows, as the models see them (dd_cluster.py:212-216)
    c3 = [X3[:, i] for i in range(14)]
    outs3 = {}
    for which in ("ddtheta", "ddgamma"):
        fns = [sympy.lambdify(syms3, sympy.sympify(r["sympy_format"], locals=dict(zip(names3, syms3))), "numpy")
               for r in g3[which]["rows"]]
        with np.errstate(all="ignore"):
            outs3[which] = np.stack([np.broadcast_to(np.asarray(f(*c3), float), (128,)) for f in fns])
    np.savez(f"{OUT}/kat_dynamics_gen3.npz", X=X3, out_theta=outs3["ddtheta"], out_gamma=outs3["ddgamma"])
    # features_dd itself (savgol window 11 / order 3, np.gradient chains, surge / sway) on a synthetic log
    Td = 160
    td = np.cumsum(rng3.uniform(0.03, 0.07, Td))
    dfd = pd.DataFrame({"Time": td})
    P0d = 1000 * np.stack([0.1 * np.sin(0.3 * td), 0.1 * np.cos(0.2 * td), 0.05 * td], 1) + rng3.normal(0, 2, (Td, 3))
    P1d = P0d + 1000 * np.stack([0.6 + 0.2 * np.sin(0.5 * td), -0.4 + 0.1 * td, 0.3 * np.cos(0.4 * td)], 1)
    Vd = np.stack([80 * np.sin(0.7 * td), 30 * np.cos(0.5 * td), 20 * np.sin(0.9 * td + 1)], 1) + rng3.normal(0, 3, (Td, 3))
    for nm, arr in (("rod_end", P0d), ("robot_cable_attach_point", P1d), ("rob_cor_speed", Vd)):
        for j, ax in enumerate("XYZ"):
            dfd[f"{nm} {ax}"] = arr[:, j]
    dfd["Theta"] = 0.3 * np.sin(0.8 * td) + rng3.normal(0, 0.01, Td)
    dfd["Gamma"] = -0.2 * np.cos(0.6 * td) + rng3.normal(0, 0.01, Td)
    Fd, Yd = main_fun.features_dd(dfd)
    np.savez(f"{OUT}/kat_features_dd.npz", time=td, P0=P0d, P1=P1d, V=Vd, theta=dfd["Theta"].values, gamma=dfd["Gamma"].values,
             features=Fd, targets=Yd)

    # 12. smoothing helpers of the analysis scripts on the same log: preprocess_signals (gaussian_filter1d, sigma 2,
    #     main_fun.py:768-776) and compute_derivatives (savgol 11/3 + two np.gradient passes, main_fun.py:645-655)
    tt, th_s, ga_s = main_fun.preprocess_signals(dfd, sigma=2)
    tt3, th_s3, ga_s3 = main_fun.preprocess_signals(dfd.iloc[:6], sigma=3.5)         # log shorter than the kernel radius
    ddt, ddg = main_fun.compute_derivatives(dfd)
    # ---- Lagrangian evaluation: fixtures the reference itself holds (outputs of its Lagrangian runs: the trajectory the
    # residuals were evaluated on, the residual series, the expression texts), plus synthetic Lagrangians through the
    # reference's route (sympy diff / solve / lambdify, restated in the oracle) --------------------------------------------
    from oracle import rovmpc_oracle as orc_l
    lag = {}
    for tag, run in (("full", "Lg_C6_full_1K_20250424_130306"), ("split_hy", "Lg_C6_split_Hy_1K_20it_20250424_165101"),
                     ("split", "Lg_C6_split_1K_20it_20250424_151011")):
        d = os.path.join(REF, "outputs", run)
        tr = np.load(os.path.join(d, "trajectory_data.npz"), allow_pickle=False)
        rs = np.load(os.path.join(d, "euler_lagrange_residuals.npz"), allow_pickle=False)
        txt = open(os.path.join(d, "best_lagrangian.txt" if tag == "full" else "lagrangian_expression.txt")).read().strip()
        for k in ("theta", "gamma", "dtheta", "dgamma", "ddtheta", "ddgamma", "time"):
            if tag == "full":
                lag[k] = tr[k]
            else:
                assert np.array_equal(tr[k], lag[k])                  # the three runs share one training trajectory
        lag[f"expr_{tag}"] = np.array(txt)
        lag[f"residual_theta_{tag}"] = rs["residual_theta"].astype(np.float64)
        lag[f"residual_gamma_{tag}"] = rs["residual_gamma"].astype(np.float64)
        # the oracle's restatement reproduces the reference's stored residuals
        r_th, r_ga = orc_l.el_residuals(txt, *(lag[k] for k in ("theta", "gamma", "dtheta", "dgamma", "ddtheta", "ddgamma")))
        assert np.allclose(r_th, lag[f"residual_theta_{tag}"], rtol=1e-12, atol=1e-15), tag
        assert np.allclose(r_ga, lag[f"residual_gamma_{tag}"], rtol=1e-12, atol=1e-15), tag
    synth = ["0.5*x2**2 + 0.5*sin(x0)**2*x3**2 + 9.81*cos(x0) - 0.3*x1**2",
             "x2**2*(1.0 + 0.2*cos(x1)) + 0.7*x3**2*exp(-x0**2) - tanh(x0*x1) - 0.5*x1**2",
             "(x2*x2 + x3*x3)*sqrt(1.5 + x0*x0) - Abs(x0) - square(x1) + x0*x3/(2.0 + x1*x1)"]
    rs_ = np.random.default_rng(77)
    rows = rs_.normal(0.0, 0.6, size=(300, 6))
    lag["synth_rows"] = rows
    lag["synth_exprs"] = np.array(synth)
    tsyn = np.cumsum(np.r_[0.0, rs_.uniform(0.004, 0.012, 199)])
    lag["synth_time"] = tsyn
    y0s = rs_.normal(0.0, 0.3, size=(5, 4))
    lag["synth_y0"] = y0s
    for j, txt in enumerate(synth):
        r_th, r_ga = orc_l.el_residuals(txt, *rows.T)
        lag[f"synth_res_theta_{j}"] = r_th; lag[f"synth_res_gamma_{j}"] = r_ga
        if j < 2:                                                    # the third couples ddtheta and ddgamma: not isolable
            f_th, f_ga = orc_l.lagrangian_accelerations(txt)
            lag[f"synth_rollout_{j}"] = np.array([orc_l.lagrangian_rollout(f_th, f_ga, tsyn, *y) for y in y0s])   # (5, 4, T)
    np.savez(f"{OUT}/kat_lagrangian.npz", **lag)

    np.savez(f"{OUT}/kat_smoothing.npz", theta_gauss2=th_s, gamma_gauss2=ga_s, theta_gauss35_first6=th_s3, gamma_gauss35_first6=ga_s3,
             ddtheta=ddt, ddgamma=ddg)

    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
