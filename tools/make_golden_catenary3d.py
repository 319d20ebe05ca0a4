#!/usr/bin/env python3
"""Golden vectors for the reference's own in-repo catenary generator (build container only; needs /root/reference).

* ``compute_catenary_3D`` (models/catenary_3d.py:5-39) is pulled out of its script with ``ast`` and exec'd here: the
  module itself opens an interactive matplotlib window at import (its last line), the function is self-contained;
* ``main_fun.transform_catenary`` (imported) is run with that function as its ``catenary_fn`` -- the whole
  augmented-catenary path on reference code alone, no stand-in for the absent pympc class.

Writes tests/golden/kat_catenary_3d.npz and tests/golden/kat_transform_catenary_3d.npz (a separate script so that the
fixtures of tools/make_golden.py stay byte for byte what they were)."""
import os
import sys
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")
from make_golden import extract_function  # noqa: E402


def main():
    ns = {"np": np}
    exec(extract_function(f"{REF}/models/catenary_3d.py", "compute_catenary_3D"), ns)
    ref_fn = ns["compute_catenary_3D"]
    import main_fun
    rng = np.random.default_rng(20250607)
    M = 16
    p0s, p1s, ropes, pts = [], [], [], []
    cases = [((0.0, 0.0, 10.0), (10.0, 0.0, 10.0), 12.0),          # the script's own demo (models/catenary_3d.py:48-50)
             ((0.0, 0.0, 0.0), (0.2435, -0.7583, 0.2980), 3.0),    # the scaler's mean attach point, 3 m cable
             ((0.0, 0.0, 0.0), (0.0, 0.0, 2.0), 3.0),              # vertical pair
             ((0.0, 0.0, 0.0), (2.0, 2.0, 1.0), 3.0),              # exactly taut: |p1 - p0| = 3
             ((0.0, 0.0, 0.0), (3.0, 1.0, 0.5), 3.0),              # rope shorter than the distance
             ((0.1, -0.2, 0.3), (0.1 + 2.9, -0.2, 0.3), 3.0)]      # nearly taut: many rounds of the fixed point
    for _ in range(34):
        a = rng.uniform(-0.3, 0.3, 3)
        b = a + rng.uniform(-1.6, 1.6, 3)
        cases.append((tuple(a), tuple(b), float(rng.choice([3.0, 3.0, 3.0, 2.0, 5.0]))))
    for a, b, L in cases:
        p0s.append(a); p1s.append(b); ropes.append(L)
        pts.append(ref_fn(np.array(a, float), np.array(b, float), L, M))
    np.savez(f"{OUT}/kat_catenary_3d.npz", p0=np.array(p0s, float), p1=np.array(p1s, float), rope=np.array(ropes),
             points=np.array(pts), M=M)

    cat = lambda s, e: (None, None, None, ref_fn(np.asarray(s, float), np.asarray(e, float), 3.0, M))   # noqa: E731
    A, B, th, ga, outs = [], [], [], [], [[], [], [], []]
    for i in range(30):
        a = np.zeros(3) if i % 3 else rng.uniform(-0.3, 0.3, 3)
        b = a + (rng.uniform(-1.5, 1.5, 3) if i != 7 else np.array([0.0, 0.0, 1.7]))      # one vertical connection
        if i == 11:
            b = a + np.array([2.5, 1.5, 1.0])                                              # beyond the cable: straight
        t, g = np.radians(rng.uniform(-20, 20)), np.radians(rng.uniform(-30, 30))
        r = main_fun.transform_catenary(a.copy(), b.copy(), cat, t, g)
        A.append(a); B.append(b); th.append(t); ga.append(g)
        for k in range(4):
            assert len(r[k]) == M
            outs[k].append(np.asarray(r[k], float))
    np.savez(f"{OUT}/kat_transform_catenary_3d.npz", A=np.array(A), B=np.array(B), theta=np.array(th), gamma=np.array(ga),
             original=np.array(outs[0]), theta_rotated=np.array(outs[1]), theta_aligned=np.array(outs[2]),
             final=np.array(outs[3]), L=3.0, M=M)
    print("wrote kat_catenary_3d.npz, kat_transform_catenary_3d.npz")


if __name__ == "__main__":
    main()
