"""R model-formula front end: parse `y ~ expr`, recognise registered device models.

The reference builds the model closure from the formula's right-hand side,
`.fn <- function(par, .data) eval(formula[[3]], c(as.list(par), .data))`
(R/nls.R:565), and a Jacobian closure with stats::deriv (R/nls.R:588-599).  A GPU
cannot evaluate R closures, so this module is the "model lowering" step of the
drop-in boundary: the RHS is parsed into a small AST and matched
structurally (up to renaming of parameters / data columns) against the formulas of the
device row-model registry (gslnls_amd/csrc/models.hpp).

The AST also evaluates with numpy, which the tests use to hand the *oracle* the very
same model as Python callbacks.
"""
import math
import re

import numpy as np

_TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)|([A-Za-z_.][A-Za-z_.0-9]*)|(\*\*|<=|>=|==|!=|[-+*/^()~,<>]))")

def _pnorm(x):
    from scipy.special import erfc
    return 0.5 * erfc(-np.asarray(x, dtype=np.float64) * 0.70710678118654752440)


def _polygamma(x, n=0):
    from scipy.special import digamma, polygamma
    n = int(n)
    return digamma(x) if n == 0 else polygamma(n, x)


def _gammafn(name):
    def f(x):
        import scipy.special as sp
        return getattr(sp, name)(np.asarray(x, dtype=np.float64))
    return f


# the functions stats::deriv differentiates (R/nls.R:588-599 builds the Jacobian of a formula with it); csrc/expr_compile.hpp
# lowers the same list
FUNCS = {"exp": np.exp, "log": np.log, "sin": np.sin, "cos": np.cos, "tan": np.tan, "atan": np.arctan,
         "sqrt": np.sqrt, "abs": np.abs, "tanh": np.tanh, "sinh": np.sinh, "cosh": np.cosh, "asin": np.arcsin,
         "acos": np.arccos, "log1p": np.log1p, "expm1": np.expm1, "log2": np.log2, "log10": np.log10, "pnorm": _pnorm,
         "dnorm": lambda x: np.exp(-0.5 * np.asarray(x, dtype=np.float64) ** 2) * 0.39894228040143267794,
         "sinpi": lambda x: np.sin(np.pi * np.asarray(x)), "cospi": lambda x: np.cos(np.pi * np.asarray(x)),
         "tanpi": lambda x: np.tan(np.pi * np.asarray(x)),
         "gamma": _gammafn("gamma"), "lgamma": _gammafn("gammaln"), "digamma": _gammafn("digamma"),
         "trigamma": lambda x: _polygamma(x, 1), "psigamma": _polygamma,
         "factorial": lambda x: _gammafn("gamma")(np.asarray(x, dtype=np.float64) + 1.0),
         "lfactorial": lambda x: _gammafn("gammaln")(np.asarray(x, dtype=np.float64) + 1.0),
         # R functions outside stats::deriv's table: a formula that uses them cannot be lowered to the device (nor
         # differentiated by the reference); the mirror evaluates them in the closure route, as R evaluates .fn
         "ifelse": lambda c, a, b: np.where(c, a, b), "pmax": lambda *a: np.maximum.reduce(np.broadcast_arrays(*a)),
         "pmin": lambda *a: np.minimum.reduce(np.broadcast_arrays(*a)),
         # the standard selfStart models by their closed forms (stats::SSasymp & co.)
         "SSasymp": lambda x, Asym, R0, lrc: Asym + (R0 - Asym) * np.exp(-np.exp(lrc) * x),
         "SSasympOff": lambda x, Asym, lrc, c0: Asym * (1 - np.exp(-np.exp(lrc) * (x - c0))),
         "SSasympOrig": lambda x, Asym, lrc: Asym * (1 - np.exp(-np.exp(lrc) * x)),
         "SSbiexp": lambda x, A1, lrc1, A2, lrc2: A1 * np.exp(-np.exp(lrc1) * x) + A2 * np.exp(-np.exp(lrc2) * x),
         "SSfol": lambda D, x, lKe, lKa, lCl: D * np.exp(lKe + lKa - lCl) * (np.exp(-np.exp(lKe) * x) - np.exp(-np.exp(lKa) * x))
         / (np.exp(lKa) - np.exp(lKe)),
         "SSfpl": lambda x, A, B, xmid, scal: A + (B - A) / (1 + np.exp((xmid - x) / scal)),
         "SSgompertz": lambda x, Asym, b2, b3: Asym * np.exp(-b2 * b3 ** x),
         "SSlogis": lambda x, Asym, xmid, scal: Asym / (1 + np.exp((xmid - x) / scal)),
         "SSmicmen": lambda x, Vm, K: Vm * x / (K + x),
         "SSweibull": lambda x, Asym, Drop, lrc, pwr: Asym - Drop * np.exp(-np.exp(lrc) * x ** pwr)}
CONSTS = {"pi": math.pi}


class Node:
    __slots__ = ("op", "args", "val")

    def __init__(self, op, args=(), val=None):
        self.op, self.args, self.val = op, tuple(args), val

    def __repr__(self):
        if self.op == "num":
            return repr(self.val)
        if self.op == "sym":
            return self.val
        if self.op == "call":
            return "%s(%s)" % (self.val, ", ".join(map(repr, self.args)))
        if self.op == "neg":
            return "(-%r)" % (self.args[0],)
        return "(%r %s %r)" % (self.args[0], self.op, self.args[1])


def _tokenize(s):
    pos, out = 0, []
    s = s.strip()
    while pos < len(s):
        m = _TOKEN.match(s, pos)
        if not m:
            raise ValueError("cannot tokenize formula at %r" % s[pos:pos + 10])
        num, sym, op = m.groups()
        if num is not None:
            out.append(("num", float(num)))
        elif sym is not None:
            out.append(("sym", sym))
        else:
            out.append(("op", "^" if op == "**" else op))
        pos = m.end()
    return out


class _Parser:
    # R precedence: ^ (right assoc) > unary minus > * / > + -
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def take(self):
        tok = self.peek()
        self.i += 1
        return tok

    def expr(self):
        # comparisons bind weaker than + - (R's precedence); they only occur inside ifelse() and the like
        node = self.sum()
        while self.peek()[0] == "op" and self.peek()[1] in ("<", ">", "<=", ">=", "==", "!="):
            op = self.take()[1]
            node = Node(op, (node, self.sum()))
        return node

    def sum(self):
        node = self.term()
        while self.peek() in (("op", "+"), ("op", "-")):
            op = self.take()[1]
            node = Node(op, (node, self.term()))
        return node

    def term(self):
        node = self.unary()
        while self.peek() in (("op", "*"), ("op", "/")):
            op = self.take()[1]
            node = Node(op, (node, self.unary()))
        return node

    def unary(self):
        if self.peek() == ("op", "-"):
            self.take()
            return Node("neg", (self.unary(),))
        if self.peek() == ("op", "+"):
            self.take()
            return self.unary()
        return self.power()

    def power(self):
        base = self.atom()
        if self.peek() == ("op", "^"):
            self.take()
            # exponent binds a following unary minus: x^-2
            return Node("^", (base, self.unary_pow()))
        return base

    def unary_pow(self):
        if self.peek() == ("op", "-"):
            self.take()
            return Node("neg", (self.unary_pow(),))
        return self.power()

    def atom(self):
        kind, v = self.take()
        if kind == "num":
            return Node("num", val=v)
        if kind == "sym":
            if self.peek() == ("op", "("):
                self.take()
                args = []
                if self.peek() != ("op", ")"):
                    args.append(self.expr())
                    while self.peek() == ("op", ","):
                        self.take()
                        args.append(self.expr())
                if self.take() != ("op", ")"):
                    raise ValueError("expected )")
                return Node("call", args, v)
            return Node("sym", val=v)
        if (kind, v) == ("op", "("):
            node = self.expr()
            if self.take() != ("op", ")"):
                raise ValueError("expected )")
            return node
        raise ValueError("unexpected token %r" % (v,))


def parse_expr(s):
    p = _Parser(_tokenize(s))
    node = p.expr()
    if p.i != len(p.t):
        raise ValueError("trailing tokens in %r" % s)
    return node


def parse_formula(s):
    """'lhs ~ rhs' -> (lhs_node or None, rhs_node)"""
    if "~" not in s:
        raise ValueError("formula needs '~'")
    lhs, rhs = s.split("~", 1)
    lhs = lhs.strip()
    return (parse_expr(lhs) if lhs else None), parse_expr(rhs)


def symbols(node, acc=None):
    acc = [] if acc is None else acc
    if node.op == "sym":
        if node.val not in acc and node.val not in CONSTS:
            acc.append(node.val)
    for a in node.args:
        symbols(a, acc)
    return acc


def evaluate(node, env):
    op = node.op
    if op == "num":
        return node.val
    if op == "sym":
        if node.val in env:
            return env[node.val]
        if node.val in CONSTS:
            return CONSTS[node.val]
        raise KeyError(node.val)
    if op == "neg":
        return -evaluate(node.args[0], env)
    if op == "call":
        return FUNCS[node.val](*[evaluate(a, env) for a in node.args])
    a, b = evaluate(node.args[0], env), evaluate(node.args[1], env)
    if op == "+":
        return a + b
    if op == "-":
        return a - b
    if op == "*":
        return a * b
    if op == "/":
        return a / b
    if op == "^":
        return np.power(a, b)
    cmp = {"<": np.less, ">": np.greater, "<=": np.less_equal, ">=": np.greater_equal, "==": np.equal, "!=": np.not_equal}
    if op in cmp:
        return cmp[op](a, b)
    raise ValueError(op)


def _match(t, u, pmap, dmap, tparams, uparams):
    """structural match of template t against user AST u with consistent symbol renaming"""
    if t.op != u.op or len(t.args) != len(u.args):
        return False
    if t.op == "num":
        return t.val == u.val
    if t.op == "sym":
        t_is_par, u_is_par = t.val in tparams, u.val in uparams
        if t_is_par != u_is_par:
            return False
        m = pmap if t_is_par else dmap
        if t.val in m:
            return m[t.val] == u.val
        if u.val in m.values():
            return False
        m[t.val] = u.val
        return True
    if t.op == "call" and t.val != u.val:
        return False
    return all(_match(a, b, pmap, dmap, tparams, uparams) for a, b in zip(t.args, u.args))


# registry: id -> (template formula RHS, parameter names in device order, regressor names)
REGISTRY = {
    1: ("A*exp(-lam*x)+b", ("A", "lam", "b"), ("x",)),
    2: ("b1*(1-exp(-b2*x))", ("b1", "b2"), ("x",)),
    3: ("a*exp(-(x-b)^2/(2*c^2))", ("a", "b", "c"), ("x",)),
    4: ("b1*exp(-b2*x) + b3*exp(-(x-b4)^2/b5^2) + b6*exp(-(x-b7)^2/b8^2)",
        ("b1", "b2", "b3", "b4", "b5", "b6", "b7", "b8"), ("x",)),
}


def lower_c(rhs_text, param_names):
    """The same lowering through the C ABI (gslnls_lower_formula, csrc/formula.hpp) -- what the R shim calls."""
    import ctypes as C
    from . import _lib
    p = len(param_names)
    arr = (C.c_char_p * p)(*[s.encode() for s in param_names])
    order = (C.c_int * p)()
    buf = C.create_string_buffer(256)
    mid = _lib.lib().gslnls_lower_formula(rhs_text.encode(), p, arr, order, buf, 256)
    if mid <= 0:
        return None
    return mid, list(order), buf.value.decode().split(",")


def lower(rhs, param_names):
    """Find the registered device model for AST `rhs`.

    Returns (model_id, order, xnames): order[k] = index into param_names of the k-th device
    parameter, xnames = data column names in device order.  None when nothing matches.
    """
    uparams = set(param_names)
    for mid, (tmpl, tpar, tx) in REGISTRY.items():
        if len(tpar) != len(param_names):
            continue
        pmap, dmap = {}, {}
        if _match(parse_expr(tmpl), rhs, pmap, dmap, set(tpar), uparams):
            if len(pmap) != len(tpar):
                continue
            order = [list(param_names).index(pmap[t]) for t in tpar]
            return mid, order, [dmap[t] for t in tx]
    return None
