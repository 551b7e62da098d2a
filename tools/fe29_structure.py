"""fe29_structure.py -- ties the interval proofs of tools/radix29_model.py to the DEVICE code they are about.

radix29_model.py restates the functions of ecsimd_amd/csrc/fe29.cuh by hand ("same names, same order"); its exact model is checked against big-int
formulas and the device against the oracle, but until round 5 nothing checked model == device STRUCTURALLY: a carry pass (norm29) or value
reduction (vred29) dropped from the device and not from the model would have passed every test while voiding the overflow proof.

This module builds, for each function, the expression DAG of its outputs twice --

  * from the C++ source: a small parser for the statement forms fe29.cuh uses (declarations, assignments to locals / members / reference
    parameters, `if constexpr (...)` on the curve's traits, the one ternary on TIGHT_SQ, cswap29, calls of other fe29 functions), with the
    primitives sqr29 / mul29 / add29 / sub29 / dbl29 / norm29<S> / vred29 as DAG nodes;
  * from the model: the same Python functions the proofs execute, run on a symbolic executor (`Sym`) instead of intervals --

and compares them node for node (multiplication and addition are commutative: their operands are ordered canonically on both sides).
tests/test_radix29_model.py asserts equality for every function and curve the proofs cover, and that deleting any single norm29 / vred29 from
the device source is detected.
"""
import os
import re

import radix29_model as m

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FE29 = os.path.join(ROOT, "ecsimd_amd", "csrc", "fe29.cuh")

# what `r29_prime<C>::xxx` evaluates to for each curve the proofs cover (fe29.cuh r29_prime)
TRAITS = {
    "p256": {"p256": True, "dense": False, "tight_sq": True},
    "secp256k1": {"p256": False, "dense": False, "tight_sq": False},
    "generic": {"p256": False, "dense": True, "tight_sq": True},
}
MODEL_CURVE = {"p256": m.CURVE_P256, "secp256k1": m.CURVE_SECP, "generic": m.CURVE_ANY}


def _canon(op, *args):
    if op in ("mul", "add"):
        args = tuple(sorted(args, key=repr))
    return (op,) + tuple(args)


# ---------------------------------------------------------------------------------------------------------------- the model, symbolically
class Sym:
    """The executor interface of radix29_model (Exact / Bounds) on symbols: every method returns the DAG node of its result."""
    def __init__(self, curve):
        self.cv = curve

    def add(self, a, b): return _canon("add", a, b)
    def sub(self, a, b): return _canon("sub", a, b)
    def dbl(self, a): return _canon("dbl", a)
    def norm(self, a, shift=0): return _canon("norm", a, shift)
    def mul(self, a, b): return _canon("mul", a, b)
    def sqr(self, a): return _canon("sqr", a)
    def vred(self, a): return _canon("vred", a)
    def cswap(self, sw, a, b): return _canon("cswap", 0, a, b), _canon("cswap", 1, a, b)


# ---------------------------------------------------------------------------------------------------------------- the device source, parsed
def _strip_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def function_source(name, text=None):
    """(parameter list, body) of the fe29.cuh function `name` (the first definition whose name matches exactly)."""
    text = _strip_comments(text if text is not None else open(FE29).read())
    mo = re.search(r"ECS_DEV\s+[\w:<>]+\s+" + re.escape(name) + r"\s*\(", text)
    if not mo:
        raise KeyError(name)
    i = mo.end()
    depth, j = 1, i
    while depth:
        depth += {"(": 1, ")": -1}.get(text[j], 0); j += 1
    params = text[i:j - 1]
    k = text.index("{", j)
    depth, e = 1, k + 1
    while depth:
        depth += {"{": 1, "}": -1}.get(text[e], 0); e += 1
    return params, text[k + 1:e - 1]


def _split_top(s, sep):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[{" and not (ch == "<" and False):
            depth += ch in "([{"
        if ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    out.append(cur)
    return [x.strip() for x in out]


class Device:
    """Interprets one fe29.cuh function on symbols."""
    PRIMS = {"sqr29": "sqr", "mul29": "mul", "add29": "add", "sub29": "sub", "dbl29": "dbl", "vred29": "vred"}

    def __init__(self, kind, text=None, flags=None):
        self.kind, self.traits, self.text = kind, TRAITS[kind], text
        self.flags = dict(flags or {})                     # template booleans such as NOZ

    # ---- expressions
    def expr(self, s, env):
        s = s.strip()
        while s.startswith("(") and self._matching(s, 0) == len(s) - 1:
            s = s[1:-1].strip()
        q = self._find_top(s, "?")
        if q >= 0:                                           # COND ? a : b with a compile-time COND
            c = self._find_top(s, ":")
            return self.expr(s[q + 1:c], env) if self.cond(s[:q]) else self.expr(s[c + 1:], env)
        mo = re.match(r"^([A-Za-z_]\w*)\s*(<[^()]*>)?\s*\(", s)
        if mo and self._matching(s, mo.end() - 1) == len(s) - 1:
            name, targs = mo.group(1), (mo.group(2) or "")
            args = _split_top(s[mo.end():-1], ",")
            args = [a for a in args if a not in ("cx", "")]           # the per-curve context rides along everywhere: not a value
            if name in self.PRIMS:
                return _canon(self.PRIMS[name], *[self.expr(a, env) for a in args])
            if name == "norm29":
                sh = int(targs.strip("<>")) if targs.strip("<>").strip().isdigit() else 0
                return _canon("norm", self.expr(args[0], env), sh)
            return self.call(name, [self.ref_or_val(a, env) for a in args], env)
        if re.fullmatch(r"[A-Za-z_][\w.]*", s):
            if s not in env:
                raise KeyError(f"unknown name {s!r}")
            return env[s]
        raise ValueError(f"cannot parse expression {s!r}")

    def ref_or_val(self, a, env):
        a = a.strip()
        return ("name", a) if re.fullmatch(r"[A-Za-z_][\w.]*", a) else ("val", self.expr(a, env))

    def cond(self, c):
        c = c.strip()
        neg = c.startswith("!")
        c = c.lstrip("!").strip()
        mo = re.fullmatch(r"r29_prime<C>::(\w+)", c)
        if mo:
            v = self.traits[mo.group(1)]
        elif c in self.flags:
            v = self.flags[c]
        else:
            raise ValueError(f"unknown compile-time condition {c!r}")
        return (not v) if neg else v

    @staticmethod
    def _matching(s, i):
        depth = 0
        for j in range(i, len(s)):
            depth += {"(": 1, ")": -1}.get(s[j], 0)
            if depth == 0:
                return j
        return -1

    @staticmethod
    def _find_top(s, ch):
        depth = 0
        for j, c in enumerate(s):
            if c in "(<":
                depth += 1 if c == "(" else 0
            elif c == ")":
                depth -= 1
            elif c == ch and depth == 0 and not (ch == ":" and s[j:j + 2] == "::") and not (ch == ":" and j and s[j - 1] == ":"):
                return j
        return -1

    # ---- statements
    def call(self, name, args, caller_env):
        """Runs fe29.cuh function `name`; args are ("name", caller's name) or ("val", node).  Returns the return value; reference parameters are
        written back into the caller's environment."""
        params, body = function_source(name, self.text)
        plist = [p for p in _split_top(params, ",") if p and "r29_ctx" not in p]
        env, refs = {}, []
        for p, a in zip(plist, args):
            pname = re.search(r"(\w+)\s*(=.*)?$", p).group(1)
            by_ref = "&" in p and not p.strip().startswith("const")
            if re.match(r"\s*(const\s+)?(coz29|jpoint29)\b", p):            # a struct: bind its members
                src = a[1]
                for k, v in list(caller_env.items()):
                    if k.startswith(src + "."):
                        env[pname + k[len(src):]] = v
                if by_ref:
                    refs.append((pname, src, True))
            elif re.match(r"\s*(const\s+)?fe29\b", p):
                if a[0] == "name" and a[1] in caller_env:
                    env[pname] = caller_env[a[1]]
                elif a[0] == "val":
                    env[pname] = a[1]
                if by_ref:
                    refs.append((pname, a[1], False))
            else:                                                          # uint32_t masks and the like: opaque
                env[pname] = ("arg", pname)
        ret = self.block(body, env)
        for pname, src, is_struct in refs:
            if is_struct:
                for k, v in env.items():
                    if k.startswith(pname + "."):
                        caller_env[src + k[len(pname):]] = v
            elif pname in env:
                caller_env[src] = env[pname]
        return ret

    def block(self, body, env):
        i, n = 0, len(body)
        while i < n:
            while i < n and body[i] in " \t\n":
                i += 1
            if i >= n:
                break
            if body.startswith("if constexpr", i):
                j = body.index("(", i)
                k = self._matching(body, j)
                c = self.cond(body[j + 1:k])
                then, after = self._braced(body, k + 1)
                els = None
                rest = body[after:].lstrip()
                if rest.startswith("else"):
                    off = after + (len(body[after:]) - len(rest)) + 4
                    els, after = self._braced(body, off)
                r = self.block(then if c else (els or ""), env)
                if r is not None:
                    return r
                i = after
                continue
            j = i
            depth = 0
            while j < n and not (body[j] == ";" and depth == 0):
                depth += {"(": 1, ")": -1, "{": 1, "}": -1}.get(body[j], 0); j += 1
            st = body[i:j].strip()
            i = j + 1
            if not st or st.startswith("#pragma"):
                continue
            r = self.statement(st, env)
            if r is not None:
                return r
        return None

    def _braced(self, body, i):
        """(content, index after) of the `{...}` block or single statement that starts at or after i."""
        while body[i] in " \t\n":
            i += 1
        if body[i] == "{":
            depth, e = 1, i + 1
            while depth:
                depth += {"{": 1, "}": -1}.get(body[e], 0); e += 1
            return body[i + 1:e - 1], e
        e = body.index(";", i)
        return body[i:e + 1], e + 1

    def statement(self, st, env):
        if st.startswith("return"):
            v = st[6:].strip()
            if re.fullmatch(r"\w+", v) and (v + ".x") in env:                # a struct local
                return {k[len(v) + 1:]: val for k, val in env.items() if k.startswith(v + ".")}
            return self.expr(v, env)
        if st.startswith("constexpr bool"):
            mo = re.match(r"constexpr bool (\w+)\s*=\s*(.*)$", st)
            self.flags[mo.group(1)] = self.cond(mo.group(2))
            return None
        mo = re.match(r"^cswap29\((\w+),\s*(\w+),\s*(\w+)\)$", st)
        if mo:
            a, b = env[mo.group(2)], env[mo.group(3)]
            env[mo.group(2)], env[mo.group(3)] = _canon("cswap", 0, a, b), _canon("cswap", 1, a, b)
            return None
        st = re.sub(r"^(const\s+)?(fe29|jpoint29|coz29)\s+", "", st)        # a declaration is an assignment (or nothing) here
        if "=" in st:
            for decl in _split_top(st, ","):                                 # `const fe29 Yn = norm29(P.y), Zn = norm29(P.z);`
                lhs, rhs = decl.split("=", 1)
                val = self.expr(rhs, env)
                if isinstance(val, dict):                                    # a struct returned by value: bind its members
                    for k, v in val.items():
                        env[lhs.strip() + "." + k] = v
                else:
                    env[lhs.strip()] = val
            return None
        if re.fullmatch(r"[\w\s,]+", st):                                   # `fe29 H, r;` / `jpoint29 R;`
            return None
        mo = re.match(r"^([A-Za-z_]\w*)\s*(<[^()]*>)?\s*\((.*)\)$", st, re.S)
        if mo:                                                               # a call for its reference parameters
            self.call(mo.group(1), [self.ref_or_val(a, env) for a in _split_top(mo.group(3), ",") if a not in ("cx", "")], env)
            return None
        raise ValueError(f"cannot parse statement {st!r}")


# ---------------------------------------------------------------------------------------------------------------- the comparisons
def _syms(*names):
    return {n: ("in", n) for n in names}


def zdau(kind, text=None, noz=False):
    """(device DAGs, model DAGs) of the ladder iteration's six outputs."""
    inp = _syms("x1", "x2", "dx", "y1", "dy", "z")
    d = Device(kind, text, {"NOZ": noz})
    env = {"s." + k: v for k, v in inp.items()}
    env["oswap"] = ("arg", "oswap")
    d.call("zdau29", [("name", "s"), ("name", "oswap")], env)
    dev = {k: env["s." + k] for k in inp}
    mod = m.zdau29(Sym(MODEL_CURVE[kind]), dict(inp), True)
    if noz:
        mod["z"] = inp["z"]
    return dev, mod


def _point(prefix, names=("x", "y", "z")):
    return {prefix + "." + n: ("in", prefix + "." + n) for n in names}


def madd(kind, text=None):
    env = dict(_point("P"), x2=("in", "x2"), y2=("in", "y2"))
    r = Device(kind, text).call("madd29", [("name", "P"), ("name", "x2"), ("name", "y2")], env)
    X, Y, Z = m.madd29(Sym(MODEL_CURVE[kind]), env["P.x"], env["P.y"], env["P.z"], env["x2"], env["y2"])
    return r, {"x": X, "y": Y, "z": Z}


def maddv(kind, text=None):
    env = dict(_point("P"), x2=("in", "x2"), y2=("in", "y2"))
    d = Device(kind, text)
    d.call("madd29_hr", [("name", "P"), ("name", "x2"), ("name", "y2"), ("name", "H"), ("name", "r")], env)
    r = d.call("madd29v_finish", [("name", "P"), ("name", "H"), ("name", "r")], env)
    X, Y, Z, H, rr = m.madd29v(Sym(MODEL_CURVE[kind]), env["P.x"], env["P.y"], env["P.z"], env["x2"], env["y2"], with_hr=True)
    return dict(r, H=env["H"], r=env["r"]), {"x": X, "y": Y, "z": Z, "H": H, "r": rr}


def jdbl(kind, text=None):
    env = _point("P")
    r = Device(kind, text).call("jdbl29", [("name", "P")], env)
    X, Y, Z = m.jdbl29(Sym(MODEL_CURVE[kind]), env["P.x"], env["P.y"], env["P.z"])
    return r, {"x": X, "y": Y, "z": Z}


def dbl_add(kind, text=None):
    env = dict(_point("P"), x2=("in", "x2"), y2=("in", "y2"))
    r = Device(kind, text).call("dbl_add29", [("name", "P"), ("name", "x2"), ("name", "y2")], env)
    X, Y, Z = m.dbl_add29(Sym(MODEL_CURVE[kind]), env["P.x"], env["P.y"], env["P.z"], env["x2"], env["y2"])
    return r, {"x": X, "y": Y, "z": Z}


def zaddu(kind, text=None):
    env = _syms("x1", "y1", "x2", "y2", "z")
    Device(kind, text).call("zaddu29", [("name", n) for n in ("x1", "y1", "x2", "y2", "z", "rx", "ry", "dx")], env)
    inp = _syms("x1", "y1", "x2", "y2", "z")
    rx, ry, W1, A1, z, dx = m.zaddu29(Sym(MODEL_CURVE[kind]), inp["x1"], inp["y1"], inp["x2"], inp["y2"], inp["z"])
    return {k: env[k] for k in ("rx", "ry", "x1", "y1", "z", "dx")}, {"rx": rx, "ry": ry, "x1": W1, "y1": A1, "z": z, "dx": dx}


def pdbl(kind, text=None):
    env = _point("P")
    r = Device(kind, text).call("pdbl29", [("name", "P")], env)
    X, Y, Z = m.pdbl29(Sym(MODEL_CURVE[kind]), env["P.x"], env["P.y"], env["P.z"])
    return r, {"x": X, "y": Y, "z": Z}


def padd(kind, text=None):
    env = dict(_point("P"), x2=("in", "x2"), y2=("in", "y2"))
    r = Device(kind, text).call("padd29", [("name", "P"), ("name", "x2"), ("name", "y2")], env)
    X, Y, Z = m.padd29(Sym(MODEL_CURVE[kind]), env["P.x"], env["P.y"], env["P.z"], env["x2"], env["y2"])
    return r, {"x": X, "y": Y, "z": Z}


def gjdbl(kind, text=None, wout=True):
    """fe29.cuh gjdbl29<C, WOUT>: the point and the carried w = a Z^4 (a reference parameter)."""
    env = dict(_point("P"), w=("in", "w"))
    r = Device(kind, text, {"WOUT": wout}).call("gjdbl29", [("name", "P"), ("name", "w")], env)
    X, Y, Z, W3 = m.gjdbl29(Sym(MODEL_CURVE[kind]), env["P.x"], env["P.y"], env["P.z"], ("in", "w"), wout)
    return dict(r, w=env["w"]), {"x": X, "y": Y, "z": Z, "w": W3}


def gjdbl_last(kind, text=None):
    return gjdbl(kind, text, wout=False)


# every (function, curve) pair an interval proof of radix29_model.py speaks about
COVERED = [(zdau, "p256"), (zdau, "secp256k1"), (zdau, "generic"), (madd, "p256"), (madd, "secp256k1"), (jdbl, "p256"), (jdbl, "secp256k1"),
           (dbl_add, "p256"), (dbl_add, "secp256k1"), (maddv, "p256"), (maddv, "secp256k1"), (zaddu, "secp256k1"), (pdbl, "secp256k1"), (padd, "secp256k1"),
           (gjdbl, "generic"), (gjdbl_last, "generic"), (dbl_add, "generic"), (zaddu, "generic")]      # a registered curve's window loop and table (k_gvarwin.hip)


def count(node, op):
    """How many DISTINCT nodes of kind `op` the DAG(s) contain."""
    seen = set()

    def walk(x):
        if isinstance(x, dict):
            for v in x.values():
                walk(v)
        elif isinstance(x, tuple) and x and isinstance(x[0], str):
            if x in seen:
                return
            seen.add(x)
            for a in x[1:]:
                walk(a)
    walk(node)
    return sum(1 for x in seen if x[0] == op)


if __name__ == "__main__":
    for fn, kind in COVERED:
        dev, mod = fn(kind)
        same = dev == mod
        print(f"{fn.__name__:8s} {kind:10s} {'device == model' if same else 'DIFFERENT'}: {count(dev, 'mul')} products, {count(dev, 'sqr')} squares, "
              f"{count(dev, 'norm')} carry passes, {count(dev, 'vred')} value reductions")
        if not same:
            for k in dev:
                if dev[k] != mod.get(k):
                    print("   output", k, "differs")
