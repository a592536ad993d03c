"""Front-end for the leaf strings of the reference.

`fabrics` configures every leaf with two Python expression strings in `x`, `xdot` and `ca.*`
(example_pandas_Jointspace.py:87-89, example_pointmasses_static.py:106-107).  The device code
evaluates two closed families (include/mrf.h, mrf_leaf_fn):

    POW      :  k / x**p * gate(xdot) * xdot**2
    LOGISTIC :  k * (1/(1 + c*exp(-s*x)) - 1) * gate(xdot) * xdot**2

with gate in {1, [xdot<0]} -- `(1 - ca.heaviside(xdot))` and `-0.5*(ca.sign(xdot) - 1)` are the same
function (1, 0.5, 0 for xdot <, =, > 0).  `parse_leaf` reduces a string to that family structurally
and then *verifies* the result numerically against a direct evaluation of the string; anything
outside the families raises LeafSpecError instead of being approximated.
"""
import ast
import math
import random

from . import abi


class LeafSpecError(ValueError):
    pass


# ------------------------------------------------------------------ tiny expression tree
class _N:
    def __init__(self, op, *args, val=None):
        self.op, self.args, self.val = op, args, val

    def __repr__(self):
        return f"{self.op}({', '.join(map(repr, self.args))})" if self.args else f"{self.op}:{self.val}"


def _const(v):
    return _N("const", val=float(v))


_FUNCS = {"exp", "sign", "heaviside", "fabs", "log", "tanh", "sqrt"}


def _build(node):
    if isinstance(node, ast.Expression):
        return _build(node.body)
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return _const(node.value)
    if isinstance(node, ast.Name):
        if node.id in ("x", "xdot"):
            return _N(node.id)
        raise LeafSpecError(f"unknown name {node.id!r}")
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        a = _build(node.operand)
        return a if isinstance(node.op, ast.UAdd) else _N("mul", _const(-1.0), a)
    if isinstance(node, ast.BinOp):
        a, b = _build(node.left), _build(node.right)
        if isinstance(node.op, ast.Add):
            return _N("add", a, b)
        if isinstance(node.op, ast.Sub):
            return _N("add", a, _N("mul", _const(-1.0), b))
        if isinstance(node.op, ast.Mult):
            return _N("mul", a, b)
        if isinstance(node.op, ast.Div):
            return _N("mul", a, _N("pow", b, _const(-1.0)))
        if isinstance(node.op, ast.Pow):
            return _N("pow", a, b)
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name) \
            and node.func.value.id in ("ca", "np", "math") and node.func.attr in _FUNCS and len(node.args) == 1:
        return _N(node.func.attr, _build(node.args[0]))
    raise LeafSpecError(f"unsupported syntax: {ast.dump(node)[:80]}")


def _eval(n, x, xd):
    o = n.op
    if o == "const":
        return n.val
    if o == "x":
        return x
    if o == "xdot":
        return xd
    if o == "add":
        return _eval(n.args[0], x, xd) + _eval(n.args[1], x, xd)
    if o == "mul":
        return _eval(n.args[0], x, xd) * _eval(n.args[1], x, xd)
    if o == "pow":
        return _eval(n.args[0], x, xd) ** _eval(n.args[1], x, xd)
    a = _eval(n.args[0], x, xd)
    if o == "exp":
        return math.exp(a)
    if o == "sign":
        return (a > 0) - (a < 0)
    if o == "heaviside":
        return 0.5 * (((a > 0) - (a < 0)) + 1.0)
    if o == "fabs":
        return abs(a)
    if o == "log":
        return math.log(a)
    if o == "tanh":
        return math.tanh(a)
    if o == "sqrt":
        return math.sqrt(a)
    raise LeafSpecError(o)


def _depends(n, name):
    return n.op == name or any(_depends(a, name) for a in n.args)


def _is_const(n):
    return not _depends(n, "x") and not _depends(n, "xdot")


def _flatten_mul(n, expo=1.0, out=None):
    """n == coef * prod(factor ** e): returns (coef, [(factor, e), ...])."""
    if out is None:
        out = []
    coef = 1.0
    if _is_const(n):
        return _eval(n, 0.0, 0.0) ** expo, out
    if n.op == "mul":
        for a in n.args:
            c, _ = _flatten_mul(a, expo, out)
            coef *= c
        return coef, out
    if n.op == "pow" and _is_const(n.args[1]):
        e = _eval(n.args[1], 0.0, 0.0)
        c, _ = _flatten_mul(n.args[0], expo * e, out)
        return c, out
    out.append((n, expo))
    return coef, out


def _flatten_add(n, out=None):
    if out is None:
        out = []
    if n.op == "add":
        for a in n.args:
            _flatten_add(a, out)
    else:
        out.append(n)
    return out


def _match_gate(n):
    """A function of xdot alone that is c*[xdot<0] (with c/2 at 0): returns c, else None."""
    if _depends(n, "x"):
        return None
    neg = [_eval(n, 0.0, v) for v in (-3.0, -1.0, -1e-3)]
    pos = [_eval(n, 0.0, v) for v in (1e-3, 1.0, 3.0)]
    if max(neg) - min(neg) > 1e-14 or any(abs(v) > 1e-14 for v in pos) or abs(neg[0]) < 1e-300:
        return None
    if abs(_eval(n, 0.0, 0.0) - 0.5 * neg[0]) > 1e-14:
        return None
    return neg[0]


def _match_logistic(n):
    """n == k*(1/(1 + c*exp(-s*x)) - 1): returns (k, c, s), else None."""
    if _depends(n, "xdot"):
        return None
    const, others = 0.0, []
    for t in _flatten_add(n):
        if _is_const(t):
            const += _eval(t, 0.0, 0.0)
        else:
            others.append(t)
    if len(others) != 1:
        return None
    a, facs = _flatten_mul(others[0])
    if len(facs) != 1 or facs[0][1] != -1.0 or facs[0][0].op != "add":
        return None
    b, inner = 0.0, []
    for t in _flatten_add(facs[0][0]):
        if _is_const(t):
            b += _eval(t, 0.0, 0.0)
        else:
            inner.append(t)
    if len(inner) != 1 or b == 0.0:
        return None
    cc, ifacs = _flatten_mul(inner[0])
    if len(ifacs) != 1 or ifacs[0][1] != 1.0 or ifacs[0][0].op != "exp":
        return None
    ss, afacs = _flatten_mul(ifacs[0][0].args[0])
    if len(afacs) != 1 or afacs[0][0].op != "x" or afacs[0][1] != 1.0:
        return None
    k = a / b
    if abs(const + k) > 1e-12 * max(1.0, abs(k)):
        return None
    return k, cc / b, -ss


def parse_leaf(expr):
    """Reduce a leaf string to an abi.LeafFn; raise LeafSpecError outside the two families."""
    try:
        tree = _build(ast.parse(expr.strip(), mode="eval"))
    except SyntaxError as e:
        raise LeafSpecError(f"cannot parse leaf string {expr!r}: {e}") from None
    coef, facs = _flatten_mul(tree)
    xpow = xdpow = 0.0
    gate, logistic = abi.GATE_NONE, None
    for f, e in facs:
        if f.op == "x":
            xpow += e
        elif f.op == "xdot":
            xdpow += e
        else:
            g = _match_gate(f) if e == 1.0 else None
            lg = _match_logistic(f) if e == 1.0 else None
            if g is not None and gate == abi.GATE_NONE:
                gate, coef = abi.GATE_NEG, coef * g
            elif lg is not None and logistic is None:
                logistic = lg
            else:
                raise LeafSpecError(f"leaf string {expr!r}: factor {f!r}**{e} is outside the supported families")
    if xdpow != 2.0:
        raise LeafSpecError(f"leaf string {expr!r}: must be homogeneous of degree 2 in xdot (got {xdpow})")
    fn = abi.LeafFn()
    fn.gate = gate
    if logistic is not None:
        if xpow != 0.0:
            raise LeafSpecError(f"leaf string {expr!r}: logistic factor times a power of x is not supported")
        fn.family, fn.p = abi.FAMILY_LOGISTIC, 0
        fn.k, fn.c, fn.s = coef * logistic[0], logistic[1], logistic[2]
    else:
        p = -xpow
        if p != int(p) or not (0 <= p <= 16):
            raise LeafSpecError(f"leaf string {expr!r}: x exponent {-p} must be an integer power 1/x**p, 0<=p<=16")
        fn.family, fn.p = abi.FAMILY_POW, int(p)
        fn.k, fn.c, fn.s = coef, 0.0, 0.0
    # verification against the string itself
    rng = random.Random(1234)
    for _ in range(64):
        x = rng.uniform(0.02, 3.0)
        xd = rng.choice((-1.0, 1.0)) * rng.uniform(1e-3, 3.0)
        want = _eval(tree, x, xd)
        got = evaluate(fn, x, xd)
        if abs(want - got) > 1e-12 * max(1.0, abs(want)):
            raise LeafSpecError(f"leaf string {expr!r}: family fit disagrees with the string at x={x}, xdot={xd}")
    return fn


def leaf_coeff(fn, x, xd):
    g = 1.0 if fn.gate == abi.GATE_NONE else (1.0 if xd < 0 else (0.0 if xd > 0 else 0.5))
    if fn.family == abi.FAMILY_POW:
        return fn.k / x ** fn.p * g
    return fn.k * (1.0 / (1.0 + fn.c * math.exp(-fn.s * x)) - 1.0) * g


def evaluate(fn, x, xd):
    """Value of the leaf string (geometry h, or Finsler energy L) at (x, xdot)."""
    return leaf_coeff(fn, x, xd) * xd * xd


def metric(fn, x, xd):
    """d2L/dxdot2 of a Finsler string."""
    return 2.0 * leaf_coeff(fn, x, xd)
