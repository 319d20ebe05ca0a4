import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def scaler():
    s = json.load(open(os.path.join(GOLDEN, "scaler.json")))
    return np.array(s["mean"]), np.array(s["scale"])


@pytest.fixture(scope="session")
def equations():
    return json.load(open(os.path.join(GOLDEN, "equations.json")))


def chosen_row(eqs, which):
    c = eqs[which]["chosen_complexity"]
    for r in eqs[which]["rows"]:
        if r["complexity"] == c:
            return r
    raise KeyError(c)


@pytest.fixture(scope="session")
def oracle_model(scaler, equations):
    from oracle import rovmpc_oracle as orc
    mean, scale = scaler
    return orc.DynamicsModel(mean, scale,
                             orc.SymbolicModel(chosen_row(equations, "dtheta_dt")["sympy_format"]),
                             orc.SymbolicModel(chosen_row(equations, "dgamma_dt")["sympy_format"]))
