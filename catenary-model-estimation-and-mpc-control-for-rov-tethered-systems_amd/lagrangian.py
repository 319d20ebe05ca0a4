"""Euler-Lagrange residuals and the second-order rollout of a discovered Lagrangian.

Mirrors the evaluation half of the reference's Lagrangian runs -- NOT the discovery (PySR) half:

* ``compute_EL_residuals`` / ``evaluate`` (lagrangian_pipeline_old.py:60-90; lagrangian_pipeline.py:130-174,
  LagrangianModelEstimator.py:158-195): for L(theta, gamma, dtheta, dgamma) given as an expression over x0..x3,
      EOM_q = d/dt (dL/d dq) - dL/dq,   d/dt f = f_theta dtheta + f_gamma dgamma + f_dtheta ddtheta + f_dgamma ddgamma,
  evaluated on every row of a trajectory (theta, gamma, dtheta, dgamma, ddtheta, ddgamma);
* the explicit accelerations ``sp.solve(EOM_theta, ddtheta)[0]`` / ``sp.solve(EOM_gamma, ddgamma)[0]``
  (lagrangian_pipeline.py:146-171) and the forward integration of evaluate_lagrangian_on_test.py:59-68.

The symbolic part (differentiation, collecting the acceleration coefficients) is done here on the expression tree of
``expr.py``'s grammar -- no computer-algebra dependency in the product -- and compiled to the same bytecode the rollout
kernel's interpreter runs; the numeric part (residual rows, rollouts) runs on the GPU through ``rovmpc_eval_expression`` and
``rovmpc_lagrangian_rollout``.
"""
from __future__ import annotations

import ast
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .expr import ExpressionError, Program, compile_expression

# variables of the derived expressions: x0..x5 = theta, gamma, dtheta, dgamma, ddtheta, ddgamma
Q_NAMES = ("x0", "x1", "x2", "x3", "x4", "x5")

_FUNCS = {"sin", "cos", "tanh", "exp", "log", "sqrt", "abs", "Abs", "square", "neg"}


# ---- a small expression algebra on Python ast nodes ---------------------------------------------------------------------

def _num(v: float) -> ast.AST:
    return ast.Constant(float(v))


def _is_num(n, v=None) -> bool:
    if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)) and not isinstance(n.value, bool):
        return v is None or float(n.value) == v
    return False


def _add(a, b):
    if _is_num(a, 0.0):
        return b
    if _is_num(b, 0.0):
        return a
    if _is_num(a) and _is_num(b):
        return _num(a.value + b.value)
    return ast.BinOp(a, ast.Add(), b)


def _sub(a, b):
    if _is_num(b, 0.0):
        return a
    if _is_num(a, 0.0):
        return _neg(b)
    if _is_num(a) and _is_num(b):
        return _num(a.value - b.value)
    return ast.BinOp(a, ast.Sub(), b)


def _neg(a):
    if _is_num(a):
        return _num(-a.value)
    if isinstance(a, ast.UnaryOp) and isinstance(a.op, ast.USub):
        return a.operand
    return ast.UnaryOp(ast.USub(), a)


def _mul(a, b):
    if _is_num(a, 0.0) or _is_num(b, 0.0):
        return _num(0.0)
    if _is_num(a, 1.0):
        return b
    if _is_num(b, 1.0):
        return a
    if _is_num(a) and _is_num(b):
        return _num(a.value * b.value)
    return ast.BinOp(a, ast.Mult(), b)


def _div(a, b):
    if _is_num(a, 0.0):
        return _num(0.0)
    if _is_num(b, 1.0):
        return a
    return ast.BinOp(a, ast.Div(), b)


def _pow(a, e: float):
    if e == 0.0:
        return _num(1.0)
    if e == 1.0:
        return a
    return ast.BinOp(a, ast.Pow(), _num(e))


def _call(fn: str, a):
    return ast.Call(ast.Name(fn, ast.Load()), [a], [])


def _const_exponent(node) -> Optional[float]:
    if _is_num(node):
        return float(node.value)
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub) and _is_num(node.operand):
        return -float(node.operand.value)
    return None


def differentiate(node: ast.AST, var: str) -> ast.AST:
    """d node / d var on the grammar of ``expr.py`` (+ - * / **, unary minus, sin cos tanh exp log sqrt abs square neg)."""
    if _is_num(node):
        return _num(0.0)
    if isinstance(node, ast.Name):
        return _num(1.0 if node.id == var else 0.0)
    if isinstance(node, ast.UnaryOp):
        d = differentiate(node.operand, var)
        return _neg(d) if isinstance(node.op, ast.USub) else d
    if isinstance(node, ast.BinOp):
        a, b = node.left, node.right
        if isinstance(node.op, ast.Add):
            return _add(differentiate(a, var), differentiate(b, var))
        if isinstance(node.op, ast.Sub):
            return _sub(differentiate(a, var), differentiate(b, var))
        if isinstance(node.op, ast.Mult):
            return _add(_mul(differentiate(a, var), b), _mul(a, differentiate(b, var)))
        if isinstance(node.op, ast.Div):
            da, db = differentiate(a, var), differentiate(b, var)
            if _is_num(db, 0.0):
                return _div(da, b)
            return _div(_sub(_mul(da, b), _mul(a, db)), _mul(b, b))
        if isinstance(node.op, ast.Pow):
            e = _const_exponent(b)
            if e is None:
                raise ExpressionError("only constant exponents can be differentiated")
            return _mul(_mul(_num(e), _pow(a, e - 1.0)), differentiate(a, var))
        raise ExpressionError(f"unsupported operator {type(node.op).__name__}")
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and len(node.args) == 1 and node.func.id in _FUNCS:
        fn, u = node.func.id, node.args[0]
        du = differentiate(u, var)
        if _is_num(du, 0.0):
            return _num(0.0)
        if fn == "sin":
            return _mul(_call("cos", u), du)
        if fn == "cos":
            return _mul(_neg(_call("sin", u)), du)
        if fn == "tanh":
            return _mul(_sub(_num(1.0), _call("square", _call("tanh", u))), du)
        if fn == "exp":
            return _mul(_call("exp", u), du)
        if fn == "log":
            return _div(du, u)
        if fn == "sqrt":
            return _div(du, _mul(_num(2.0), _call("sqrt", u)))
        if fn in ("abs", "Abs"):
            # sign(u), as sympy differentiates Abs for real u -- sign(0) = 0 there, so not u / |u| (0 / 0): the denominator is
            # |u| + 1e-300, which leaves every quotient with |u| >= 1e-284 unchanged to the last bit
            return _mul(_div(u, _add(_call("abs", u), ast.Constant(1e-300))), du)
        if fn == "square":
            return _mul(_mul(_num(2.0), u), du)
        if fn == "neg":
            return _neg(du)
    raise ExpressionError(f"cannot differentiate: {ast.dump(node)[:60]}")


def _depends_on(node: ast.AST, var: str) -> bool:
    return any(isinstance(n, ast.Name) and n.id == var for n in ast.walk(node))


def _text(node: ast.AST) -> str:
    return ast.unparse(ast.fix_missing_locations(ast.Expression(node)))


def _parse(text: str, variable_names: Optional[Sequence[str]]) -> ast.AST:
    src = text.strip().replace("^", "**")
    try:
        tree = ast.parse(src, mode="eval").body
    except SyntaxError as e:
        raise ExpressionError(f"cannot parse expression {text!r}: {e}") from None
    if variable_names:                                             # e.g. the reference's symbols theta, gamma, dtheta, dgamma
        ren = {n: Q_NAMES[i] for i, n in enumerate(variable_names)}

        class R(ast.NodeTransformer):
            def visit_Name(self, n):
                return ast.Name(ren.get(n.id, n.id), ast.Load())
        tree = R().visit(tree)
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and n.id not in Q_NAMES[:4] and n.id not in _FUNCS and n.id not in ("pow", "Pow"):
            raise ExpressionError(f"a Lagrangian is a function of x0..x3 (theta, gamma, dtheta, dgamma); found {n.id!r}")
    return tree


@dataclass
class EulerLagrange:
    """Expressions over x0..x5 = (theta, gamma, dtheta, dgamma, ddtheta, ddgamma)."""
    lagrangian: str
    eom_theta: str            # residual of the theta equation (lagrangian_pipeline_old.py:66-73)
    eom_gamma: str            # ... of the gamma equation (:76-83)
    acc_theta: Optional[str]  # ddtheta solved from eom_theta = 0 over x0..x3 (lagrangian_pipeline.py:146-152); None if not isolable
    acc_gamma: Optional[str]


def euler_lagrange(lagrangian: str, variable_names: Optional[Sequence[str]] = None) -> EulerLagrange:
    L = _parse(lagrangian, variable_names)
    th, ga, dth, dga, ddth, ddga = (ast.Name(n, ast.Load()) for n in Q_NAMES)

    def eom(q: str, dq: str):
        p = differentiate(L, dq)                                   # dL/d dq
        # total time derivative along the trajectory (:67-72): the acceleration coefficients are kept apart
        a_th, a_ga = differentiate(p, "x2"), differentiate(p, "x3")
        rest = _sub(_add(_mul(differentiate(p, "x0"), dth), _mul(differentiate(p, "x1"), dga)), differentiate(L, q))
        full = _add(_add(_mul(a_th, ddth), _mul(a_ga, ddga)), rest)
        return full, a_th, a_ga, rest

    e_th, a_tt, a_tg, r_th = eom("x0", "x2")
    e_ga, a_gt, a_gg, r_ga = eom("x1", "x3")

    def solve(own, cross, rest):
        # sp.solve(EOM, ddq)[0] = -(rest + cross * ddq_other) / own; the reference lambdifies it over (theta, gamma, dtheta,
        # dgamma) only, so a surviving cross term means "not isolable" there too.  Whether a coefficient vanishes is decided
        # as sympy's simplification would: syntactically, or -- for one that is zero only after like terms cancel --
        # numerically on sample points.  A vanishing own coefficient: sp.solve returns [] and the reference's pipeline
        # (lagrangian_pipeline.py:150-153, 168-171) catches the IndexError and integrates with zero acceleration; here that is
        # the string "0.0" with `vanishing` recorded, and lagrangian_rollout(on_unsolvable=...) decides what to do with it.
        if not (_is_num(cross, 0.0) or _vanishes(cross)):
            return None, False
        if _is_num(own, 0.0) or _vanishes(own):
            return "0.0", True
        return _text(_div(_neg(rest), own)), False

    (acc_t, van_t), (acc_g, van_g) = solve(a_tt, a_tg, r_th), solve(a_gg, a_gt, r_ga)
    el = EulerLagrange(_text(L), _text(e_th), _text(e_ga), acc_t, acc_g)
    el.vanishing = (van_t, van_g)
    return el



def _vanishes(node: ast.AST) -> bool:
    """True when the expression over x0..x5 evaluates to (numerically) zero on sample points although it is not the literal 0:
    a coefficient such as x0 - x0 or sin(x1)**2 + cos(x1)**2 - 1, which sympy's solve would cancel."""
    rng = np.random.default_rng(12345)
    pts = rng.uniform(-2.0, 2.0, size=(6, 64))
    ns = {"sin": np.sin, "cos": np.cos, "tanh": np.tanh, "exp": np.exp, "log": lambda v: np.log(np.abs(v) + 1e-300),
          "sqrt": lambda v: np.sqrt(np.abs(v)), "abs": np.abs, "Abs": np.abs, "square": np.square, "neg": np.negative}
    ns.update({f"x{i}": pts[i] for i in range(6)})
    try:
        with np.errstate(all="ignore"):
            v = np.asarray(eval(compile(ast.Expression(ast.fix_missing_locations(node)), "<coefficient>", "eval"), {"__builtins__": {}}, ns),
                           dtype=np.float64)
    except Exception:                                     # noqa: BLE001 -- anything odd: treat as non-zero (the syntactic rule stands)
        return False
    v = np.broadcast_to(v, (64,)) if v.ndim == 0 else v
    return bool(np.all(np.isfinite(v)) and np.max(np.abs(v)) < 1e-12)


def _compile(text: str, n_features: int) -> Tuple[Program, List[float]]:
    consts: List[float] = []
    prog = compile_expression(text, consts, n_features)
    return prog, consts


def el_residuals(lagrangian: str, theta, gamma, dtheta, dgamma, ddtheta, ddgamma,
                 variable_names: Optional[Sequence[str]] = None, engine=None):
    """(residual_theta, residual_gamma) on every trajectory row: ``evaluate`` of lagrangian_pipeline_old.py:85-90."""
    from .engine import default_engine
    eng = engine or default_engine()
    el = euler_lagrange(lagrangian, variable_names)
    X = np.ascontiguousarray(np.column_stack([np.asarray(v, np.float64).reshape(-1) for v in
                                              (theta, gamma, dtheta, dgamma, ddtheta, ddgamma)]))
    out = []
    for text in (el.eom_theta, el.eom_gamma):
        prog, consts = _compile(text, 6)
        out.append(eng.eval_expression(prog.code, consts, X))
    return out[0], out[1]


def lagrangian_rollout(lagrangian: str, time, theta0, gamma0, vtheta0, vgamma0,
                       variable_names: Optional[Sequence[str]] = None, engine=None, on_unsolvable: str = "raise"):
    """Forward integration of evaluate_lagrangian_on_test.py:59-68 with the accelerations solved from the Lagrangian:
    a = dd(theta_{i-1}, gamma_{i-1}, v_{i-1});  v_i = v_{i-1} + a dt;  q_i = q_{i-1} + v_{i-1} dt.
    The initial values may be arrays of equal length B: B independent rollouts in one launch.  Returns (theta, gamma,
    vtheta, vgamma), each (T,) or (B, T)."""
    from .engine import default_engine
    eng = engine or default_engine()
    el = euler_lagrange(lagrangian, variable_names)
    if on_unsolvable not in ("raise", "zero"):
        raise ValueError("on_unsolvable must be 'raise' or 'zero'")
    if any(getattr(el, "vanishing", (False, False))) and on_unsolvable == "raise":
        raise ExpressionError("the Euler-Lagrange equations of this Lagrangian cannot be solved for ddtheta / ddgamma: an acceleration "
                              "coefficient vanishes (sp.solve returns []); on_unsolvable='zero' integrates with zero acceleration, as "
                              "the reference's pipeline does after catching the IndexError (lagrangian_pipeline.py:150-153)")
    if el.acc_theta is None or el.acc_gamma is None:
        raise ExpressionError("the Euler-Lagrange equations of this Lagrangian cannot be solved for ddtheta / ddgamma separately "
                              "(vanishing or coupled acceleration terms): the reference's sp.solve(...)[0] path fails on it too")
    consts: List[float] = []
    pt = compile_expression(el.acc_theta, consts, 4)
    pg = compile_expression(el.acc_gamma, consts, 4)
    y0 = np.stack(np.broadcast_arrays(*(np.asarray(v, np.float64) for v in (theta0, gamma0, vtheta0, vgamma0))), axis=-1)
    scalar = y0.ndim == 1
    res = eng.lagrangian_rollout(pt.code, pg.code, consts, np.asarray(time, np.float64), y0.reshape(-1, 4))
    return tuple(r[0] if scalar else r for r in res)
