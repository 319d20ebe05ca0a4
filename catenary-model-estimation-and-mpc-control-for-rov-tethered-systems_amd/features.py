"""Feature map of generation 1 (simply.py:15-41; main_fun.py:167-193 is the same without the
two ``_prev`` columns).  ``extract_features`` / ``extract_features_arrays`` run on the GPU
(``rovmpc_extract_features``); there is no host restatement in the product (the oracle holds one, for
the tests).  Inside the rollout the same rows are built per (candidate, node) by the fused kernel."""
from __future__ import annotations

import numpy as np


def extract_features_arrays(P0, P1, V1, time, theta, gamma, with_prev: bool = True) -> np.ndarray:
    """P0, P1 in metres (the reference divides its mm columns by 1000 first), V1 raw.  GPU."""
    from .engine import default_engine
    return default_engine().extract_features(P0, P1, V1, time, theta, gamma, with_prev)


def extract_features(df, with_prev: bool = True) -> np.ndarray:
    """Same column names as the reference's data frames (simply.py:16-19)."""
    P0 = df[["rod_end X", "rod_end Y", "rod_end Z"]].values / 1000
    P1 = df[["robot_cable_attach_point X", "robot_cable_attach_point Y", "robot_cable_attach_point Z"]].values / 1000
    V1 = df[["rob_cor_speed X", "rob_cor_speed Y", "rob_cor_speed Z"]].values
    return extract_features_arrays(P0, P1, V1, df["Time"].values, df["Theta"].values, df["Gamma"].values, with_prev)


def features_dd_arrays(P0_mm, P1_mm, V_mm, time, theta, gamma, window: int = 11, polyorder: int = 3):
    """features_dd on arrays as logged (mm, mm/s).  GPU.  Returns (features (T,14), targets (T,2))."""
    from .engine import default_engine
    return default_engine().features_dd(P0_mm, P1_mm, V_mm, time, theta, gamma, window, polyorder)


def features_dd(df):
    """Same signature and column names as main_fun.py:811 -- returns (features, targets) of the second-order runs."""
    P0 = df[["rod_end X", "rod_end Y", "rod_end Z"]].values
    P1 = df[["robot_cable_attach_point X", "robot_cable_attach_point Y", "robot_cable_attach_point Z"]].values
    V1 = df[["rob_cor_speed X", "rob_cor_speed Y", "rob_cor_speed Z"]].values
    return features_dd_arrays(P0, P1, V1, df["Time"].values, df["Theta"].values, df["Gamma"].values)


def preprocess_signals(df, sigma=2):
    """main_fun.py:768-776: (time, gaussian-smoothed Theta, gaussian-smoothed Gamma).  GPU."""
    from .engine import default_engine
    e = default_engine()
    return df["Time"].values, e.gaussian_filter1d(df["Theta"].values, sigma), e.gaussian_filter1d(df["Gamma"].values, sigma)


def compute_derivatives(df):
    """main_fun.py:645-655: (ddtheta, ddgamma) = second np.gradient of the Savitzky-Golay (11, 3) smoothed angles.  GPU."""
    T = len(df["Time"].values)
    z = np.zeros((T, 3))
    _, Y = features_dd_arrays(z, z + 1.0, z, df["Time"].values, df["Theta"].values, df["Gamma"].values)
    return Y[:, 0], Y[:, 1]
