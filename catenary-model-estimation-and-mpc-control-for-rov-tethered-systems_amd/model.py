"""Learned dynamics artefacts: the StandardScaler constants and the chosen rows of the PySR
Pareto fronts (saved_models/scaler.pkl, equations_d{theta,gamma}_dt.csv, eq_*.txt)."""
from __future__ import annotations

import csv
import json
import os
import re
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from .expr import Program, compile_expression

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# convert.py:12-20 -- the authoritative feature-index -> name map of generation 1
FEATURE_NAMES_GEN1 = ["P1x", "P1y", "P1z", "V1x", "V1y", "V1z", "A1x", "A1y", "A1z",
                      "unit_rel_x", "unit_rel_y", "unit_rel_z", "tension", "angle_proj",
                      "theta", "gamma", "theta_prev", "gamma_prev"]


@dataclass
class DynamicsModel:
    """Scaler + the two compiled expressions; what ``rovmpc_set_model`` takes."""
    mean: np.ndarray
    scale: np.ndarray
    expr_theta: str
    expr_gamma: str
    consts: List[float] = field(default_factory=list)
    prog_theta: Optional[Program] = None
    prog_gamma: Optional[Program] = None
    variable_names: Optional[Sequence[str]] = None     # PySR variable_names (dd_cluster.py:160-168); x0..xN always work

    def __post_init__(self):
        self.mean = np.ascontiguousarray(self.mean, dtype=np.float64)
        self.scale = np.ascontiguousarray(self.scale, dtype=np.float64)
        if self.mean.shape != self.scale.shape or self.mean.ndim != 1:
            raise ValueError("mean and scale must be 1-D arrays of equal length")
        n = self.n_features
        self.consts = []
        self.prog_theta = compile_expression(self.expr_theta, self.consts, n, self.variable_names)
        self.prog_gamma = compile_expression(self.expr_gamma, self.consts, n, self.variable_names)

    @property
    def n_features(self) -> int:
        return int(self.mean.shape[0])


def read_equation_csv(path: str):
    """Rows of a saved_models/equations_*.csv (columns complexity,loss,score,equation,
    sympy_format,lambda_format) or of a PySR hall_of_fame.csv (Complexity,Loss,Equation)."""
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            r = {k.strip().lower(): v for k, v in r.items()}
            rows.append({"complexity": int(r["complexity"]), "loss": float(r["loss"]),
                         "sympy_format": r.get("sympy_format") or r["equation"],
                         "equation": r.get("equation", "")})
    return rows


def chosen_complexity_from_txt(path: str) -> int:
    """eq_*.txt is ``str(pandas.Series)`` of the chosen row (truncated with '...'); only its
    first line ``complexity   <n>`` is machine-usable (saved_models/eq_dtheta_dt.txt:1)."""
    first = open(path).readline()
    m = re.match(r"\s*complexity\s+(\d+)", first)
    if not m:
        raise ValueError(f"{path}: first line is not 'complexity <n>'")
    return int(m.group(1))


def select_row(rows, complexity: int):
    for r in rows:
        if r["complexity"] == complexity:
            return r
    raise KeyError(f"no row with complexity {complexity}")


def load_model_dir(save_dir: str, scaler_json: Optional[str] = None,
                   complexity_theta: Optional[int] = None, complexity_gamma: Optional[int] = None) -> DynamicsModel:
    """Load a ``saved_models``-style directory.  ``scaler.pkl`` is a pickle and is never
    unpickled: pass the scaler as JSON ({"mean": [...], "scale": [...]})."""
    rt = read_equation_csv(os.path.join(save_dir, "equations_dtheta_dt.csv"))
    rg = read_equation_csv(os.path.join(save_dir, "equations_dgamma_dt.csv"))
    ct = complexity_theta or chosen_complexity_from_txt(os.path.join(save_dir, "eq_dtheta_dt.txt"))
    cg = complexity_gamma or chosen_complexity_from_txt(os.path.join(save_dir, "eq_dgamma_dt.txt"))
    sj = scaler_json or os.path.join(save_dir, "scaler.json")
    s = json.load(open(sj))
    return DynamicsModel(np.array(s["mean"]), np.array(s["scale"]),
                         select_row(rt, ct)["sympy_format"], select_row(rg, cg)["sympy_format"])


def default_model(complexity_theta: Optional[int] = None, complexity_gamma: Optional[int] = None) -> DynamicsModel:
    """Generation-1 model of the reference: 18 scaled features, rows named by
    saved_models/eq_dtheta_dt.txt:1 (complexity 13) and eq_dgamma_dt.txt:1 (complexity 3)."""
    s = json.load(open(os.path.join(DATA_DIR, "gen1_scaler.json")))
    e = json.load(open(os.path.join(DATA_DIR, "gen1_equations.json")))
    ct = complexity_theta or e["dtheta_dt"]["chosen_complexity"]
    cg = complexity_gamma or e["dgamma_dt"]["chosen_complexity"]
    return DynamicsModel(np.array(s["mean"]), np.array(s["scale"]),
                         select_row(e["dtheta_dt"]["rows"], ct)["sympy_format"],
                         select_row(e["dgamma_dt"]["rows"], cg)["sympy_format"])


def generation2_model(complexity_theta: Optional[int] = None, complexity_gamma: Optional[int] = None) -> DynamicsModel:
    """Generation-2 model of the reference (what simulate_rk4_theta_gamma.py:45-46 loads): 17 UNSCALED
    features [P1, V1, A1, unit_rel, theta, gamma, cos(theta), sin(gamma), angle_proj]
    (simulate_rk4_theta_gamma.py:40), rows of outputs/differential_training_new_feature/
    d{theta,gamma}_results_20250412_163500.csv named by eq_*_20250412_163500.txt:1 (complexity 13 / 20).
    Use with ``MPCConfig(feature_map=FEATURES_GEN2)``."""
    e = json.load(open(os.path.join(DATA_DIR, "gen2_equations.json")))
    ct = complexity_theta or e["dtheta_dt"]["chosen_complexity"]
    cg = complexity_gamma or e["dgamma_dt"]["chosen_complexity"]
    m = DynamicsModel.__new__(DynamicsModel)
    m.mean = np.zeros(17); m.scale = np.ones(17)
    m.expr_theta = select_row(e["dtheta_dt"]["rows"], ct)["sympy_format"]
    m.expr_gamma = select_row(e["dgamma_dt"]["rows"], cg)["sympy_format"]
    m.__post_init__()
    return m


# dd_cluster.py:160-168 -- variable_names of the second-order runs ("gama": sympy reserves gamma)
FEATURE_NAMES_GEN3 = ["theta", "gama", "dtheta", "dgamma", "v_sway", "v_surge", "a_sway", "a_surge",
                      "V_x", "V_y", "V_z", "a_x", "a_y", "a_z"]


def generation3_model(complexity_theta: Optional[int] = None, complexity_gamma: Optional[int] = None) -> DynamicsModel:
    """Second-order model generation (dd_cluster.py): (ddtheta, ddgamma) = f(scaled features_dd row),
    14 named features (main_fun.py:849-864), rows of outputs/dd_C6_all_50_s_20250511_013928/
    d{theta,gamma}_results.csv named by eq_*.txt:1 (complexity 6 / 5), that run's scaler.
    Use with ``MPCConfig(feature_map=FEATURES_GEN3)``; ``MPCState.theta_prev / gamma_prev`` carry the
    angular rates (dtheta, dgamma) in this map."""
    s = json.load(open(os.path.join(DATA_DIR, "gen3_scaler.json")))
    e = json.load(open(os.path.join(DATA_DIR, "gen3_equations.json")))
    ct = complexity_theta or e["ddtheta"]["chosen_complexity"]
    cg = complexity_gamma or e["ddgamma"]["chosen_complexity"]
    return DynamicsModel(np.array(s["mean"]), np.array(s["scale"]),
                         select_row(e["ddtheta"]["rows"], ct)["sympy_format"],
                         select_row(e["ddgamma"]["rows"], cg)["sympy_format"],
                         variable_names=e.get("variable_names", FEATURE_NAMES_GEN3))
