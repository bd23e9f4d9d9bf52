"""Whitelisted single-expression evaluator for user formulas (host side).

Feeds already-evaluated arrays to the device path: initial conditions, the gap map and
``custom`` external generation are user expressions in the reference
(``qpsim/safe_eval.py:181-208``).  Same accepted language: one Python expression
(optionally prefixed ``return ``) over the declared variables, ``np.<whitelist>``,
``math.<whitelist>``, a few builtins and ``params.get``/``params[...]``; everything else
raises ``ValueError`` before anything is evaluated.
"""
from __future__ import annotations

import ast
import math
from typing import Any, Callable, Iterable

import numpy as np

_BUILTINS: dict[str, Callable[..., Any]] = {
    "abs": abs, "min": min, "max": max, "pow": pow, "len": len, "float": float, "int": int, "bool": bool,
}
_NP_FUNCS = frozenset(
    "abs sqrt exp log log10 sin cos tan arcsin arccos arctan sinh cosh tanh where maximum minimum clip "
    "power heaviside arange zeros_like ones_like full_like".split()
)
_NP_CONSTS = frozenset("pi e inf nan float64 float32 int64 int32 bool_".split())
_MATH_FUNCS = frozenset("sqrt exp log log10 sin cos tan asin acos atan sinh cosh tanh floor ceil".split())
_MATH_CONSTS = frozenset("pi e tau inf nan".split())
_VALUE_ATTRS = frozenset({"size", "shape"})

_STRUCTURAL = (ast.Expression, ast.BoolOp, ast.BinOp, ast.UnaryOp, ast.IfExp, ast.Compare, ast.Constant,
               ast.Slice, ast.Tuple, ast.List, ast.Dict)
_OPERATOR_TOKENS = (ast.operator, ast.unaryop, ast.boolop, ast.cmpop, ast.expr_context)


def _check_attr(node: ast.Attribute, variables: frozenset[str], *, called: bool) -> None:
    if node.attr.startswith("__"):
        raise ValueError("Dunder attribute access is not allowed in custom expressions.")
    if not isinstance(node.value, ast.Name):
        raise ValueError("Nested attribute access is not allowed in custom expressions.")
    base, attr = node.value.id, node.attr
    if base == "np":
        if attr not in (_NP_FUNCS if called else _NP_FUNCS | _NP_CONSTS):
            raise ValueError(f"Unsupported numpy attribute in custom expression: np.{attr}.")
    elif base == "math":
        if attr not in (_MATH_FUNCS if called else _MATH_FUNCS | _MATH_CONSTS):
            raise ValueError(f"Unsupported math attribute in custom expression: math.{attr}.")
    elif base == "params":
        if attr != "get":
            raise ValueError(f"Unsupported params attribute in custom expression: params.{attr}.")
    elif called:
        raise ValueError("Method calls are not allowed in custom expressions.")
    elif base in variables:
        if attr not in _VALUE_ATTRS:
            raise ValueError(f"Unsupported attribute in custom expression: {base}.{attr}.")
    else:
        raise ValueError(f"Unsupported attribute base in custom expression: {base!r}.")


def _check(node: ast.AST, variables: frozenset[str], names: frozenset[str]) -> None:
    if isinstance(node, _OPERATOR_TOKENS):
        return
    if isinstance(node, ast.Name):
        if node.id.startswith("__"):
            raise ValueError("Dunder names are not allowed in custom expressions.")
        if node.id not in names:
            raise ValueError(f"Unsupported name in custom expression: {node.id!r}.")
        return
    if isinstance(node, ast.Attribute):
        _check_attr(node, variables, called=False)
        _check(node.value, variables, names)
        return
    if isinstance(node, ast.Subscript):
        if isinstance(node.value, ast.Name) and node.value.id in ("np", "math"):
            raise ValueError("Subscript access on modules is not allowed in custom expressions.")
        _check(node.value, variables, names)
        _check(node.slice, variables, names)
        return
    if isinstance(node, ast.Call):
        if any(kw.arg is None for kw in node.keywords):
            raise ValueError("Starred keyword arguments are not allowed in custom expressions.")
        fn = node.func
        if isinstance(fn, ast.Name):
            if fn.id not in _BUILTINS:
                raise ValueError(f"Unsupported function in custom expression: {fn.id!r}.")
        elif isinstance(fn, ast.Attribute):
            if not isinstance(fn.value, ast.Name):
                raise ValueError("Nested attribute calls are not allowed in custom expressions.")
            _check_attr(fn, variables, called=True)
        else:
            raise ValueError("Unsupported call target in custom expressions.")
        _check(fn, variables, names)
        for arg in node.args:
            _check(arg, variables, names)
        for kw in node.keywords:
            _check(kw.value, variables, names)
        return
    if not isinstance(node, _STRUCTURAL):
        raise ValueError(f"Unsupported syntax in custom expression: {type(node).__name__}.")
    for child in ast.iter_child_nodes(node):
        _check(child, variables, names)


def _strip_return(source: str) -> str:
    text = str(source or "").strip()
    if not text:
        return "0.0"
    if "\n" not in text and text.startswith("return "):
        return text[len("return "):].strip()
    return text


_SHAPE_AWARE_NP = frozenset("arange zeros_like ones_like full_like".split())


def expression_is_elementwise(source: str, *, array_variables: Iterable[str]) -> bool:
    """True when ``source`` treats the variables in ``array_variables`` purely element by element: arithmetic, comparisons
    and the elementwise ``np.*`` functions only - no subscripts, ``.size`` / ``.shape``, ``len``, conditional expressions,
    ``and`` / ``or``, builtins or ``math.*`` applied to them.  Such an expression gives the same values whether it is
    evaluated bin by bin or once on broadcast arrays, which is what ``solver.evaluate_external_generation`` uses to
    evaluate a custom generation rate in ONE call.  (Syntax errors answer False; validation is not this function's job.)"""
    arrays = frozenset(array_variables)
    try:
        tree = ast.parse(_strip_return(source), mode="eval")
    except SyntaxError:
        return False

    def touches_arrays(node: ast.AST) -> bool:
        return any(isinstance(n, ast.Name) and n.id in arrays for n in ast.walk(node))

    for node in ast.walk(tree):
        if isinstance(node, (ast.IfExp, ast.BoolOp)) and touches_arrays(node):
            return False
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.Not) and touches_arrays(node):
            return False
        if isinstance(node, ast.Compare) and len(node.ops) > 1 and touches_arrays(node):
            return False       # a < x < b is an implicit `and`
        if isinstance(node, ast.Subscript) and touches_arrays(node):
            return False
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in arrays:
            return False
        if isinstance(node, (ast.Tuple, ast.List, ast.Dict)) and touches_arrays(node):
            return False
        if isinstance(node, ast.Call):
            fn = node.func
            operands = list(node.args) + [kw.value for kw in node.keywords]
            if not any(touches_arrays(a) for a in operands):
                continue
            if not (isinstance(fn, ast.Attribute) and isinstance(fn.value, ast.Name) and fn.value.id == "np"
                    and fn.attr in _NP_FUNCS - _SHAPE_AWARE_NP):
                return False
    return True


def compile_safe_expression(source: str, *, variable_names: Iterable[str]) -> Callable[..., Any]:
    """Validate ``source`` and return ``evaluate(**variables)``; all declared variables are required."""
    required = tuple(variable_names)
    try:
        tree = ast.parse(_strip_return(source), mode="eval")
    except SyntaxError as exc:
        raise ValueError(
            "Custom expressions must be a single expression (optionally prefixed by 'return ')."
        ) from exc
    variables = frozenset(required)
    _check(tree, variables, variables | frozenset(_BUILTINS) | {"np", "math"})
    code = compile(tree, "<custom-expression>", "eval")

    def evaluate(**values: Any) -> Any:
        absent = [name for name in required if name not in values]
        if absent:
            raise ValueError(f"Missing variables for custom expression evaluation: {', '.join(absent)}.")
        scope = {"__builtins__": {}, "np": np, "math": math}
        scope.update(_BUILTINS)
        scope.update(values)
        return eval(code, scope, {})  # noqa: S307 - tree was validated against the whitelist above

    return evaluate
