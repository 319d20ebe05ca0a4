"""Open-loop integrators of the reference, evaluated on the GPU.

    rk4_integration(model, x_input, time, y0)            simulate_rk4_theta_gamma.py:52-68
    integrate_theta_gamma(model_theta, model_gamma, X,
                          time_array, theta_0, gamma_0)   main_fun.py:735-764

    integrate_second_order(model_ddtheta, model_ddgamma, X, time,
                           theta_0, gamma_0, method)      test_cluster.py:110-129 ("double_euler"),
                                                          dd_cluster.py:221-226 ("trapezoid")

``model`` objects are ``SymbolicRegressor`` instances (below) -- the stand-in for the
reference's ``PySRRegressor`` restricted to one equation row: ``.predict(X(n,F)) -> (n,)``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _lib
from .engine import Engine, MPCConfig
from .model import DynamicsModel


class SymbolicRegressor:
    """One symbolic expression over (already scaled) feature rows, evaluated by librovmpc."""

    def __init__(self, expression: str, n_features: int = 18):
        self.expression = expression
        self.n_features = n_features
        self._engine: Optional[Engine] = None
        self._pair_key = None

    def _pair(self, other: "SymbolicRegressor") -> Engine:
        key = (self.expression, other.expression)
        if self._engine is None or self._pair_key != key:
            if self.n_features != other.n_features:
                raise ValueError("models disagree on n_features")
            m = DynamicsModel(np.zeros(self.n_features), np.ones(self.n_features), self.expression, other.expression)
            self._engine = Engine(MPCConfig(N=1, K=1, force_interpreter=True), m)
            self._pair_key = key
        return self._engine

    def predict(self, X) -> np.ndarray:
        return self._pair(self).predict(X, 0)

    def sympy(self) -> str:
        return self.expression


def rk4_integration(model: SymbolicRegressor, x_input, time, y0):
    th, _ = model._pair(model).replay(x_input, time, y0, y0, _lib.RK4)
    return th


def integrate_theta_gamma(model_theta: SymbolicRegressor, model_gamma: SymbolicRegressor, X, time_array,
                          theta_0, gamma_0):
    return model_theta._pair(model_gamma).replay(X, time_array, theta_0, gamma_0, _lib.EULER)


def rk4_theta_gamma(model_theta: SymbolicRegressor, model_gamma: SymbolicRegressor, X, time_array, theta_0, gamma_0):
    """Both RK4 replays of simulate_rk4_theta_gamma.py:74-75 in one launch."""
    return model_theta._pair(model_gamma).replay(X, time_array, theta_0, gamma_0, _lib.RK4)


def integrate_second_order(model_ddtheta: SymbolicRegressor, model_ddgamma: SymbolicRegressor, X, time_array,
                           theta_0, gamma_0, method: str = "double_euler"):
    """Second-derivative models (theta'', gamma''), integrated twice from zero angular velocity the
    way the reference's second-order evaluation scripts do: ``"double_euler"`` =
    test_cluster.py:110-129, ``"trapezoid"`` = cumulative_trapezoid + cumsum of
    dd_cluster.py:221-226."""
    modes = {"double_euler": _lib.DOUBLE_EULER, "trapezoid": _lib.TRAPEZOID}
    if method not in modes:
        raise ValueError("method must be 'double_euler' or 'trapezoid'")
    return model_ddtheta._pair(model_ddgamma).replay(X, time_array, theta_0, gamma_0, modes[method])
