"""
Symbolic-expression front end: text of a PySR equation row -> postfix bytecode for the HIP
interpreter (include/rovmpc.h, ``rovmpc_opcode``).

Input grammar = what the reference's equation files hold in their ``sympy_format`` /
``equation`` columns (saved_models/equations_*.csv; outputs/*/hall_of_fame*.csv):
``+ - * / **``, unary minus, numbers, variables ``x0..xN`` (or names given in
``variable_names``) and the operator vocabulary of the reference's PySR runs
(simply.py:65-66, PySRTrainingScript.py:53-54, cluster_run/train_dynamics.py:28-46,
dynamic_eq_theta_cluster.py:35-43): sin cos abs/Abs square tanh exp log sqrt neg safe_log
safe_sqrt.  Operands are emitted in Python evaluation order so the device evaluates the
same operation sequence NumPy would.
"""
from __future__ import annotations

import ast
import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

OP = {
    "PUSH_C": 0, "PUSH_F": 1, "ADD": 2, "SUB": 3, "MUL": 4, "DIV": 5, "NEG": 6, "SIN": 7, "COS": 8,
    "TANH": 9, "ABS": 10, "SQUARE": 11, "EXP": 12, "LOG": 13, "SQRT": 14, "POW": 15, "POWI": 16,
    "SAFE_LOG": 17, "SAFE_SQRT": 18,
}
MAX_CODE = 256
MAX_STACK = 16

_UNARY_CALLS = {
    "sin": "SIN", "cos": "COS", "tanh": "TANH", "exp": "EXP", "log": "LOG", "sqrt": "SQRT",
    "abs": "ABS", "Abs": "ABS", "square": "SQUARE", "neg": "NEG",
    "safe_log": "SAFE_LOG", "safe_sqrt": "SAFE_SQRT",
}
_BINOPS = {ast.Add: "ADD", ast.Sub: "SUB", ast.Mult: "MUL", ast.Div: "DIV"}


class ExpressionError(ValueError):
    pass


@dataclass
class Program:
    """Postfix program + constant pool (constants are shared by both equations of a model)."""
    code: List[int]
    max_stack: int
    features_used: List[int]
    text: str


class _Compiler:
    def __init__(self, consts: List[float], n_features: int, names: Optional[Dict[str, int]]):
        self.consts = consts
        self.n_features = n_features
        self.names = names or {}
        self.code: List[int] = []
        self.used: set = set()
        self.depth = 0
        self.max_depth = 0

    def _emit(self, op: str, arg: int = 0, delta: int = 0):
        self.code.append(((arg & 0xFFFFFF) << 8) | OP[op])
        self.depth += delta
        self.max_depth = max(self.max_depth, self.depth)

    def _const(self, v: float):
        v = float(v)
        try:
            idx = next(i for i, c in enumerate(self.consts) if c == v and np.signbit(c) == np.signbit(v))
        except StopIteration:
            idx = len(self.consts)
            self.consts.append(v)
        self._emit("PUSH_C", idx, +1)

    def _feature(self, name: str):
        if name in self.names:
            idx = self.names[name]
        else:
            m = re.fullmatch(r"x(\d+)", name)
            if not m:
                raise ExpressionError(f"unknown variable {name!r}")
            idx = int(m.group(1))
        if not 0 <= idx < self.n_features:
            raise ExpressionError(f"feature index {idx} out of range (model has {self.n_features} features)")
        self.used.add(idx)
        self._emit("PUSH_F", idx, +1)

    @staticmethod
    def _as_number(node) -> Optional[float]:
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
            return float(node.value)
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = _Compiler._as_number(node.operand)
            if v is not None:
                return -v if isinstance(node.op, ast.USub) else v
        return None

    def visit(self, node):
        num = self._as_number(node)
        if num is not None:
            self._const(num)
            return
        if isinstance(node, ast.Name):
            self._feature(node.id)
        elif isinstance(node, ast.UnaryOp):
            if isinstance(node.op, ast.UAdd):
                self.visit(node.operand)
            elif isinstance(node.op, ast.USub):
                self.visit(node.operand)
                self._emit("NEG")
            else:
                raise ExpressionError("unsupported unary operator")
        elif isinstance(node, ast.BinOp):
            if isinstance(node.op, ast.Pow):
                e = self._as_number(node.right)
                self.visit(node.left)
                if e is not None and e == 2.0:
                    self._emit("SQUARE")
                elif e is not None and float(e).is_integer() and abs(e) < (1 << 22):
                    self._emit("POWI", int(e))
                else:
                    self.visit(node.right)
                    self._emit("POW", 0, -1)
            elif type(node.op) in _BINOPS:
                self.visit(node.left)
                self.visit(node.right)
                self._emit(_BINOPS[type(node.op)], 0, -1)
            else:
                raise ExpressionError(f"unsupported operator {type(node.op).__name__}")
        elif isinstance(node, ast.Call):
            if not isinstance(node.func, ast.Name) or node.keywords:
                raise ExpressionError("unsupported call form")
            fn = node.func.id
            if fn in _UNARY_CALLS and len(node.args) == 1:
                self.visit(node.args[0])
                self._emit(_UNARY_CALLS[fn])
            elif fn in ("pow", "Pow") and len(node.args) == 2:
                self.visit(node.args[0]); self.visit(node.args[1])
                self._emit("POW", 0, -1)
            else:
                raise ExpressionError(f"unsupported function {fn!r}")
        else:
            raise ExpressionError(f"unsupported syntax: {ast.dump(node)[:60]}")


def compile_expression(text: str, consts: List[float], n_features: int = 18,
                       variable_names: Optional[Sequence[str]] = None) -> Program:
    """Compile one expression; ``consts`` is the (shared, growing) constant pool."""
    src = text.strip().replace("^", "**")
    try:
        tree = ast.parse(src, mode="eval")
    except SyntaxError as e:
        raise ExpressionError(f"cannot parse expression {text!r}: {e}") from None
    names = {n: i for i, n in enumerate(variable_names)} if variable_names else None
    c = _Compiler(consts, n_features, names)
    c.visit(tree.body)
    if len(c.code) > MAX_CODE:
        raise ExpressionError(f"expression too long ({len(c.code)} > {MAX_CODE} instructions)")
    if c.max_depth > MAX_STACK:
        raise ExpressionError(f"expression needs {c.max_depth} stack slots (max {MAX_STACK})")
    if len(consts) > MAX_CODE:
        raise ExpressionError("too many constants")
    return Program(c.code, c.max_depth, sorted(c.used), text.strip())


def disassemble(program: Program, consts: Sequence[float]) -> str:
    inv = {v: k for k, v in OP.items()}
    out = []
    for ins in program.code:
        op, arg = ins & 0xFF, ins >> 8
        name = inv[op]
        if name == "PUSH_C":
            out.append(f"PUSH_C {consts[arg]!r}")
        elif name == "PUSH_F":
            out.append(f"PUSH_F x{arg}")
        elif name == "POWI":
            out.append(f"POWI {arg - (1 << 24) if arg >= (1 << 23) else arg}")
        else:
            out.append(name)
    return "\n".join(out)
