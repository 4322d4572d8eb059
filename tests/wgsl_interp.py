"""A small interpreter for the subset of WGSL that the reference's shaders use -- TEST INFRASTRUCTURE.

Why: the two CPU restatements (oracle/orb_oracle.c, oracle/orb_numpy.py) were both written by READING the reference's shaders.  This module
EXECUTES the shader text instead (tests/test_reference_text.py reads src/shaders/*.wgsl from the reference checkout at test time -- nothing of
it is kept in this repository), so that a transcription slip in a constant, an operator, a loop bound, a bit order or a guard shows up as a
difference between three independently produced results.  It is not the reference's run time: what WGSL leaves to the implementation (the
sampler, atan2 / cos / sin, out-of-range conversions, loads outside a texture level) enters through the hooks of `Hooks`, which the test binds
to the restatement's documented decisions -- the parity claim stays "unpinned" (DESIGN.md section 2).

Supported: structs; module-scope var<private | workgroup | storage | uniform | push_constant> and const; functions; let / var; if / else;
for; compound assignment and ++; return; scalars u32 / i32 / f32 / bool with abstract literals; vecN and mat2x2 with swizzles, arithmetic and
comparisons; array constructors and indexing; atomics; the built-ins the six shaders call.  workgroupBarrier() suspends an invocation (the
statements are generators), so that a workgroup's invocations run in lock step between barriers.
Arithmetic: every binary32 operation is rounded once (NumPy float32 scalars); integers wrap to 32 bits; abstract numbers are Python int / float.
"""
import copy
import re

import numpy as np

F = np.float32
np.seterr(over="ignore", invalid="ignore", divide="ignore")


# ------------------------------------------------------------------------------------------------ values
class V:
    """A scalar: t in f32 | u32 | i32 | bool | ai (abstract int) | af (abstract float)."""
    __slots__ = ("t", "v")

    def __init__(self, t, v):
        self.t, self.v = t, v

    def __repr__(self):
        return "%s(%r)" % (self.t, self.v)


class Vec:
    """A vector of raw element values (NumPy float32 / int / bool) of element type t."""
    __slots__ = ("t", "e")

    def __init__(self, t, e):
        self.t, self.e = t, list(e)

    def __repr__(self):
        return "vec%d<%s>%r" % (len(self.e), self.t, self.e)


class Mat:
    __slots__ = ("cols",)

    def __init__(self, cols):
        self.cols = cols


class Ptr:
    """&name of a module-scope atomic."""
    __slots__ = ("scope", "name")

    def __init__(self, scope, name):
        self.scope, self.name = scope, name


class Hooks:
    """What WGSL leaves to the implementation.  The test overrides these with the restatement's decisions.
    contract_*: a shader compiler may fuse a product into the sum that consumes it -- in dot(), in matrix * vector, in
    `x += a * b`; last_first: ... and may reduce dot() / matrix * vector from the last component (Mesa's lowering)."""
    contract_dot = contract_matvec = contract_muladd = last_first = False

    def fma(self, a, b, c):
        """fl32(a * b + c), rounded once (exact rational arithmetic)."""
        from fractions import Fraction
        fr = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
        x = F(float(fr))
        cands = [np.nextafter(x, F(-np.inf)), x, np.nextafter(x, F(np.inf))]
        d = [abs(Fraction(float(q)) - fr) for q in cands]
        win = [q for q, dd in zip(cands, d) if dd == min(d)]
        return F(win[0] if len(win) == 1 else [q for q in win if (int(F(q).view(np.uint32)) & 1) == 0][0])

    def atan2(self, y, x):
        return F(np.arctan2(float(y), float(x)))

    def cos(self, a):
        return F(np.cos(float(a)))

    def sin(self, a):
        return F(np.sin(float(a)))

    def f32_to_u32(self, v):  # u32(f32): WGSL leaves values outside the range to the implementation
        v = float(v)
        return 0 if not v > 0 else min(int(v), 0xFFFFFFFF)

    def texture_load(self, tex, x, y, level):  # -> four binary32 channels
        raise NotImplementedError

    def texture_sample(self, tex, sampler, u, v):
        raise NotImplementedError

    def texture_dimensions(self, tex):
        raise NotImplementedError


def wrap(t, n):
    n &= 0xFFFFFFFF
    return n - (1 << 32) if t == "i32" and n >= (1 << 31) else n


def conv_raw(v, src, dst, hooks):
    """Raw value of type src as type dst."""
    if src == dst:
        return v
    if dst == "f32":
        return F(v)
    if dst == "af":
        return float(v)
    if dst in ("u32", "i32"):
        if src in ("f32", "af"):
            if dst == "u32":
                return hooks.f32_to_u32(v)
            f = float(v)
            return 0 if f != f else int(max(min(f, 2147483647.0), -2147483648.0))  # truncation toward zero, saturating
        return wrap(dst, int(v))
    if dst == "ai":
        return int(v)
    if dst == "bool":
        return bool(v)
    raise TypeError("conversion %s -> %s" % (src, dst))


def unify(ta, tb):
    """Common type of a binary operation's operands (abstract literals take the other side's type)."""
    if ta == tb:
        return ta
    if ta in ("ai", "af") and tb in ("ai", "af"):
        return "af"
    if ta == "ai" or (ta == "af" and tb == "f32"):
        return tb
    if tb == "ai" or (tb == "af" and ta == "f32"):
        return ta
    raise TypeError("operands of types %s and %s" % (ta, tb))


def arith(op, t, a, b):
    if t in ("f32",):
        a, b = F(a), F(b)
        if op == "+":
            return a + b
        if op == "-":
            return a - b
        if op == "*":
            return a * b
        if op == "/":
            return a / b
    elif t == "af":
        return {"+": a + b, "-": a - b, "*": a * b, "/": a / b if op == "/" else None}[op]
    elif t in ("u32", "i32", "ai"):
        if op in ("+", "-", "*"):
            r = a + b if op == "+" else (a - b if op == "-" else a * b)
        elif op == "/":
            r = 0 if b == 0 else (abs(a) // abs(b)) * (1 if (a >= 0) == (b >= 0) else -1)  # truncating division
        elif op == "%":
            r = 0 if b == 0 else a - b * ((abs(a) // abs(b)) * (1 if (a >= 0) == (b >= 0) else -1))
        elif op == "&":
            r = a & b
        elif op == "|":
            r = a | b
        elif op == "^":
            r = a ^ b
        else:
            raise TypeError(op)
        return r if t == "ai" else wrap(t, r)
    raise TypeError("operator %s on %s" % (op, t))


def compare(op, a, b):
    return {"<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b, "==": a == b, "!=": a != b}[op]


# ------------------------------------------------------------------------------------------------ lexer
TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*)
  | (?P<num>0[xX][0-9a-fA-F]+[iu]?|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?[fiu]?)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
  | (?P<op>\+\+|--|\+=|-=|\*=|/=|\|=|&=|\^=|<<=|>>=|<<|>>|<=|>=|==|!=|&&|\|\||->|[-+*/%&|^!~<>=(){}\[\],;:.@])
""", re.X)


def lex(text):
    out, pos = [], 0
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise SyntaxError("WGSL: cannot read %r" % text[pos:pos + 20])
        pos = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group()))
    out.append(("eof", ""))
    return out


def number(tok):
    if tok[:2] in ("0x", "0X"):
        suf = tok[-1] if tok[-1] in "iu" else ""
        n = int(tok[2:len(tok) - len(suf)], 16)
        return V({"u": "u32", "i": "i32", "": "ai"}[suf], n)
    suf = tok[-1] if tok[-1] in "fiu" else ""
    body = tok[:len(tok) - len(suf)]
    if suf == "f":
        return V("f32", F(float(body)))
    if suf in ("i", "u"):
        return V("i32" if suf == "i" else "u32", int(body))
    if any(c in body for c in ".eE"):
        return V("af", float(body))
    return V("ai", int(body))


# ------------------------------------------------------------------------------------------------ parser
SCALARS = {"u32", "i32", "f32", "bool"}
VEC_SUFFIX = {"i": "i32", "u": "u32", "f": "f32"}
BINARY = [["||"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="], ["<", ">", "<=", ">="], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]


class Parser:
    def __init__(self, text):
        self.t, self.i = lex(text), 0
        self.split_gt = False  # the second half of a '>>' that closed two template lists

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def accept(self, s):
        if self.peek()[1] == s and self.peek()[0] != "eof":
            self.i += 1
            return True
        return False

    def expect(self, s):
        if not self.accept(s):
            raise SyntaxError("WGSL: expected %r, found %r (token %d)" % (s, self.peek()[1], self.i))

    def close_template(self):
        if self.split_gt:
            self.split_gt = False
            return
        if self.peek()[1] == ">>":
            self.i += 1
            self.split_gt = True
            return
        self.expect(">")

    def attributes(self):
        out = {}
        while self.accept("@"):
            name = self.next()[1]
            args = []
            if self.accept("("):
                while not self.accept(")"):
                    args.append(self.next()[1])
                    self.accept(",")
            out[name] = args
        return out

    def type(self):
        name = self.next()[1]
        if name in SCALARS:
            return ("scalar", name)
        m = re.fullmatch(r"vec([234])([iuf])", name)
        if m:
            return ("vec", int(m.group(1)), VEC_SUFFIX[m.group(2)])
        m = re.fullmatch(r"vec([234])", name)
        if m:
            self.expect("<")
            el = self.type()
            self.close_template()
            return ("vec", int(m.group(1)), el[1])
        m = re.fullmatch(r"mat([234])x([234])f", name)
        if m:
            return ("mat", int(m.group(1)), int(m.group(2)), "f32")
        if name == "array":
            if self.peek()[1] != "<":
                return ("array", None, None)
            self.expect("<")
            el = self.type()
            count = None
            if self.accept(","):
                count = self.expr(no_gt=True)
            self.close_template()
            return ("array", el, count)
        if name == "atomic":
            self.expect("<")
            el = self.type()
            self.close_template()
            return ("atomic", el[1])
        if name.startswith("texture_"):
            if self.accept("<"):
                self.type()
                self.close_template()
            return ("texture",)
        if name == "sampler":
            return ("sampler",)
        return ("struct", name)

    # ---- module
    def module(self):
        structs, gvars, consts, funcs = {}, [], [], {}
        while self.peek()[0] != "eof":
            attrs = self.attributes()
            kw = self.peek()[1]
            if kw == "struct":
                self.next()
                name = self.next()[1]
                self.expect("{")
                fields = []
                while not self.accept("}"):
                    fa = self.attributes()
                    fname = self.next()[1]
                    self.expect(":")
                    fields.append((fname, self.type(), fa))
                    self.accept(",")
                self.accept(";")
                structs[name] = fields
            elif kw == "var":
                self.next()
                space = "handle"
                if self.accept("<"):
                    space = self.next()[1]
                    while self.accept(","):
                        self.next()
                    self.close_template()
                name = self.next()[1]
                ty = None
                if self.accept(":"):
                    ty = self.type()
                init = self.expr() if self.accept("=") else None
                self.expect(";")
                gvars.append((name, space, ty, init, attrs))
            elif kw == "const":
                self.next()
                name = self.next()[1]
                ty = self.type() if self.accept(":") else None
                self.expect("=")
                consts.append((name, ty, self.expr()))
                self.expect(";")
            elif kw == "fn":
                self.next()
                name = self.next()[1]
                self.expect("(")
                params = []
                while not self.accept(")"):
                    pa = self.attributes()
                    pname = self.next()[1]
                    self.expect(":")
                    params.append((pname, self.type(), pa))
                    self.accept(",")
                ret = None
                if self.accept("->"):
                    self.attributes()
                    ret = self.type()
                funcs[name] = (params, ret, self.block(), attrs)
            else:
                raise SyntaxError("WGSL: unexpected %r at module scope" % kw)
        return structs, gvars, consts, funcs

    # ---- statements
    def block(self):
        self.expect("{")
        out = []
        while not self.accept("}"):
            out.append(self.statement())
        return ("block", out)

    def simple_statement(self):
        """let / var / assignment / increment / call -- without the closing semicolon (also the clauses of `for`)."""
        kw = self.peek()[1]
        if kw in ("let", "var"):
            self.next()
            name = self.next()[1]
            ty = self.type() if self.accept(":") else None
            init = self.expr() if self.accept("=") else None
            return (kw, name, ty, init)
        lhs = self.expr()
        tok = self.peek()[1]
        if tok in ("++", "--"):
            self.next()
            return ("assign", lhs, tok[0], ("lit", V("ai", 1)))
        if tok == "=":
            self.next()
            return ("assign", lhs, None, self.expr())
        if tok in ("+=", "-=", "*=", "/=", "|=", "&=", "^=", "<<=", ">>="):
            self.next()
            return ("assign", lhs, tok[:-1], self.expr())
        return ("expr", lhs)

    def statement(self):
        kw = self.peek()[1]
        if kw == "{":
            return self.block()
        if kw == "if":
            self.next()
            cond = self.expr()
            then = self.block()
            other = None
            if self.accept("else"):
                other = self.statement() if self.peek()[1] == "if" else self.block()
            return ("if", cond, then, other)
        if kw == "for":
            self.next()
            self.expect("(")
            init = None if self.peek()[1] == ";" else self.simple_statement()
            self.expect(";")
            cond = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            step = None if self.peek()[1] == ")" else self.simple_statement()
            self.expect(")")
            return ("for", init, cond, step, self.block())
        if kw == "return":
            self.next()
            val = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            return ("return", val)
        st = self.simple_statement()
        self.expect(";")
        return st

    # ---- expressions
    def expr(self, level=0, no_gt=False):
        if level == len(BINARY):
            return self.unary()
        lhs = self.expr(level + 1, no_gt)
        while self.peek()[0] == "op" and self.peek()[1] in BINARY[level] and not (no_gt and self.peek()[1] in (">", ">>")):
            op = self.next()[1]
            lhs = ("bin", op, lhs, self.expr(level + 1, no_gt))
        return lhs

    def unary(self):
        tok = self.peek()[1]
        if tok in ("-", "!", "&") and self.peek()[0] == "op":
            self.next()
            return ("un", tok, self.unary())
        return self.postfix(self.primary())

    def postfix(self, e):
        while True:
            if self.accept("["):
                e = ("index", e, self.expr())
                self.expect("]")
            elif self.accept("."):
                e = ("member", e, self.next()[1])
            else:
                return e

    def args(self):
        out = []
        self.expect("(")
        while not self.accept(")"):
            out.append(self.expr())
            self.accept(",")
        return out

    def primary(self):
        kind, tok = self.peek()
        if kind == "num":
            self.next()
            return ("lit", number(tok))
        if tok == "(":
            self.next()
            e = self.expr()
            self.expect(")")
            return e
        if kind == "id":
            if tok in ("true", "false"):
                self.next()
                return ("lit", V("bool", tok == "true"))
            is_type = tok in SCALARS or re.fullmatch(r"vec[234][iuf]?|mat[234]x[234]f|array", tok)
            if is_type and (self.peek(1)[1] in ("(", "<")):
                ty = self.type()
                return ("construct", ty, self.args())
            self.next()
            if self.peek()[1] == "(":
                return ("call", tok, self.args())
            return ("id", tok)
        raise SyntaxError("WGSL: unexpected %r in an expression" % tok)


# ------------------------------------------------------------------------------------------------ evaluation
class Return(Exception):
    def __init__(self, value):
        self.value = value


BARRIER = "barrier"


class Module:
    def __init__(self, text, hooks=None):
        self.structs, self.gvars, self.consts, self.funcs = Parser(text).module()
        self.hooks = hooks or Hooks()
        self.globals = {}      # module scope: consts, private / storage / uniform / handle variables
        self.workgroup = {}    # var<workgroup>, reset per workgroup
        self.private_init = []
        for name, ty, e in self.consts:
            v = self.eval(e, [self.globals])
            self.globals[name] = self.coerce(v, ty) if ty else v
        for name, space, ty, init, attrs in self.gvars:
            if space == "private":
                self.private_init.append((name, ty, init))
            elif space == "workgroup":
                pass
            else:
                self.globals[name] = None  # bound by the caller
        self.reset_private()

    def reset_private(self):
        for name, ty, init in self.private_init:
            self.globals[name] = self.coerce(self.eval(init, [self.globals]), ty) if init is not None else self.zero(ty)

    def reset_workgroup(self):
        self.workgroup = {name: self.zero(ty) for name, space, ty, init, attrs in self.gvars if space == "workgroup"}

    def bind(self, **kw):
        for k, v in kw.items():
            if k not in self.globals:
                raise KeyError("the shader has no resource named %s" % k)
            self.globals[k] = v

    # ---- types
    def zero(self, ty):
        k = ty[0]
        if k == "scalar":
            return V(ty[1], F(0) if ty[1] == "f32" else (False if ty[1] == "bool" else 0))
        if k == "atomic":
            return V(ty[1], 0)
        if k == "vec":
            return Vec(ty[2], [F(0) if ty[2] == "f32" else 0] * ty[1])
        if k == "array":
            n = self.eval(ty[2], [self.globals]).v
            return [self.zero(ty[1]) for _ in range(n)]
        if k == "struct":
            return {f: self.zero(t) for f, t, _ in self.structs[ty[1]]}
        raise TypeError("no zero value of %r" % (ty,))

    def coerce(self, v, ty):
        """A value as the declared type (abstract literals become concrete)."""
        k = ty[0]
        if k == "scalar" and isinstance(v, V):
            return V(ty[1], conv_raw(v.v, v.t, ty[1], self.hooks)) if v.t in ("ai", "af") else v
        if k == "vec" and isinstance(v, Vec):
            return Vec(ty[2], [conv_raw(x, v.t, ty[2], self.hooks) for x in v.e]) if v.t in ("ai", "af") else v
        if k == "array" and isinstance(v, list) and ty[1] is not None:
            return [self.coerce(x, ty[1]) for x in v]
        return v

    def concrete(self, v):
        """let / var without a type: abstract values take their default concrete type (i32, f32)."""
        if isinstance(v, V) and v.t in ("ai", "af"):
            t = "i32" if v.t == "ai" else "f32"
            return V(t, conv_raw(v.v, v.t, t, self.hooks))
        if isinstance(v, Vec) and v.t in ("ai", "af"):
            t = "i32" if v.t == "ai" else "f32"
            return Vec(t, [conv_raw(x, v.t, t, self.hooks) for x in v.e])
        if isinstance(v, list):
            return [self.concrete(x) for x in v]
        return v

    # ---- scopes
    def lookup(self, name, scopes):
        for s in reversed(scopes):
            if name in s:
                return s
        if name in self.workgroup:
            return self.workgroup
        if name in self.globals:
            return self.globals
        raise NameError("WGSL: %s is not declared" % name)

    # ---- expressions
    def binary(self, op, a, b):
        if op in ("&&", "||"):
            raise AssertionError
        if isinstance(a, Mat) and isinstance(b, Vec) and op == "*":
            # matrix * vector = sum over the columns of column * component, every product and the sum rounded; a
            # contracting compiler fuses the later term onto the first product (CRD-13), starting from the last column if it reduces that way
            h = self.hooks
            terms = list(zip(a.cols, b.e))
            if h.last_first:
                terms.reverse()
            acc = None
            for col, x in terms:
                if acc is None:
                    acc = [arith("*", "f32", c, x) for c in col.e]
                elif h.contract_matvec:
                    acc = [h.fma(c, x, p) for c, p in zip(col.e, acc)]
                else:
                    acc = [arith("+", "f32", p, arith("*", "f32", c, x)) for c, p in zip(col.e, acc)]
            return Vec("f32", acc)
        va, vb = isinstance(a, Vec), isinstance(b, Vec)
        ta, tb = a.t, b.t
        if op in ("<<", ">>"):
            def sh(x, n):
                n &= 31
                if op == "<<":
                    return x << n if ta == "ai" else wrap(ta, x << n)
                return x >> n if ta != "u32" else (x & 0xFFFFFFFF) >> n
            if va:
                return Vec(ta, [sh(x, (b.e[i] if vb else b.v)) for i, x in enumerate(a.e)])
            return V(ta, sh(a.v, b.v))
        t = unify(ta, tb)

        def c(x, src):
            return conv_raw(x, src, t, self.hooks)
        if op in ("<", ">", "<=", ">=", "==", "!="):
            if va or vb:
                n = len(a.e) if va else len(b.e)
                return Vec("bool", [bool(compare(op, c(a.e[i] if va else a.v, ta), c(b.e[i] if vb else b.v, tb))) for i in range(n)])
            return V("bool", bool(compare(op, c(a.v, ta), c(b.v, tb))))
        if va or vb:
            n = len(a.e) if va else len(b.e)
            return Vec(t, [arith(op, t, c(a.e[i] if va else a.v, ta), c(b.e[i] if vb else b.v, tb)) for i in range(n)])
        return V(t, arith(op, t, c(a.v, ta), c(b.v, tb)))

    def construct(self, ty, args):
        k = ty[0]
        if k == "scalar":
            (a,) = args
            return V(ty[1], conv_raw(a.v, a.t, ty[1], self.hooks))
        if k == "vec":
            flat = []
            for a in args:
                if isinstance(a, Vec):
                    flat += [(x, a.t) for x in a.e]
                else:
                    flat.append((a.v, a.t))
            if len(flat) == 1 and len(args) == 1 and isinstance(args[0], V):
                flat = flat * ty[1]  # splat
            if len(flat) != ty[1]:
                raise TypeError("vec%d from %d components" % (ty[1], len(flat)))
            return Vec(ty[2], [conv_raw(x, s, ty[2], self.hooks) for x, s in flat])
        if k == "mat":
            cols, rows = ty[1], ty[2]
            flat = []
            for a in args:
                flat += [(x, a.t) for x in a.e] if isinstance(a, Vec) else [(a.v, a.t)]
            vals = [conv_raw(x, s, "f32", self.hooks) for x, s in flat]
            return Mat([Vec("f32", vals[c * rows:(c + 1) * rows]) for c in range(cols)])  # column-major
        if k == "array":
            vals = list(args)
            return [self.coerce(v, ty[1]) for v in vals] if ty[1] is not None else vals
        raise TypeError("constructor of %r" % (ty,))

    SWZ = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}

    def eval(self, e, scopes):
        k = e[0]
        if k == "lit":
            return e[1]
        if k == "id":
            return self.lookup(e[1], scopes)[e[1]]
        if k == "bin":
            op = e[1]
            if op in ("&&", "||"):
                a = self.eval(e[2], scopes)
                if (op == "&&") != a.v:
                    return V("bool", a.v)
                return V("bool", self.eval(e[3], scopes).v)
            return self.binary(op, self.eval(e[2], scopes), self.eval(e[3], scopes))
        if k == "un":
            if e[1] == "&":
                (_, name) = e[2]
                return Ptr(self.lookup(name, scopes), name)
            a = self.eval(e[2], scopes)
            if e[1] == "!":
                return V("bool", not a.v)
            neg = (lambda x, t: -x if t in ("f32", "af", "ai") else wrap(t, -x))
            return Vec(a.t, [neg(x, a.t) for x in a.e]) if isinstance(a, Vec) else V(a.t, neg(a.v, a.t))
        if k == "index":
            base, idx = self.eval(e[1], scopes), self.eval(e[2], scopes).v
            if isinstance(base, Vec):
                return V(base.t, base.e[idx])
            if isinstance(base, Mat):
                return base.cols[idx]
            if hasattr(base, "wgsl_index"):
                return base.wgsl_index(idx)
            return base[idx]
        if k == "member":
            base = self.eval(e[1], scopes)
            if isinstance(base, Vec):
                sel = [self.SWZ[c] for c in e[2]]
                return V(base.t, base.e[sel[0]]) if len(sel) == 1 else Vec(base.t, [base.e[i] for i in sel])
            return base[e[2]]
        if k == "construct":
            return self.construct(e[1], [self.eval(a, scopes) for a in e[2]])
        if k == "call":
            return self.call(e[1], [self.eval(a, scopes) for a in e[2]])
        raise TypeError("expression %r" % (k,))

    def call(self, name, args):
        h = self.hooks
        if name in self.funcs:
            params, ret, body, _ = self.funcs[name]
            scope = {p: self.coerce(a, t) for (p, t, _), a in zip(params, args)}
            try:
                for _ in self.exec(body, [scope]):
                    raise RuntimeError("a barrier inside a helper function")
            except Return as r:
                return r.value
            return None
        if name == "all":
            return V("bool", all(args[0].e))
        if name == "any":
            return V("bool", any(args[0].e))
        if name == "dot":
            # the components' products summed first to last, every operation rounded; CRD-13: fused, and / or last to first
            a, b = args
            pairs = list(zip(a.e, b.e))
            if h.last_first:
                pairs.reverse()
            acc = None
            for x, y in pairs:
                if acc is None:
                    acc = arith("*", "f32", x, y)
                elif h.contract_dot:
                    acc = h.fma(x, y, acc)
                else:
                    acc = arith("+", "f32", acc, arith("*", "f32", x, y))
            return V("f32", acc)
        if name in ("atan2", "cos", "sin"):
            vals = [conv_raw(a.v, a.t, "f32", h) for a in args]
            return V("f32", F(getattr(h, name)(*vals)))
        if name == "atomicAdd":
            p, d = args
            old = p.scope[p.name]
            p.scope[p.name] = V(old.t, wrap(old.t, old.v + d.v))
            return old
        if name == "atomicLoad":
            return args[0].scope[args[0].name]
        if name == "textureDimensions":
            w, hh = h.texture_dimensions(args[0])
            return Vec("u32", [w, hh])
        if name == "textureLoad":
            tex, xy, lvl = args
            return Vec("f32", [F(c) for c in h.texture_load(tex, int(xy.e[0]), int(xy.e[1]), int(lvl.v))])
        if name == "textureSample":
            tex, smp, uv = args
            return Vec("f32", [F(c) for c in h.texture_sample(tex, smp, F(uv.e[0]), F(uv.e[1]))])
        raise NameError("WGSL: the interpreter has no built-in %s" % name)

    # ---- statements (generators: a workgroupBarrier() yields)
    def store(self, target, value, scopes):
        k = target[0]
        value = copy.deepcopy(value) if isinstance(value, (list, dict)) else value
        if k == "id":
            scope = self.lookup(target[1], scopes)
            old = scope[target[1]]
            if isinstance(old, V) and isinstance(value, V) and value.t in ("ai", "af"):
                value = V(old.t, conv_raw(value.v, value.t, old.t, self.hooks))
            scope[target[1]] = value
        elif k == "index":
            base, idx = self.eval(target[1], scopes), self.eval(target[2], scopes).v
            if hasattr(base, "wgsl_store"):
                base.wgsl_store(idx, value)
            elif 0 <= idx < len(base):
                base[idx] = value
        elif k == "member":
            base = self.eval(target[1], scopes)
            if isinstance(base, Vec):
                base.e[self.SWZ[target[2]]] = conv_raw(value.v, value.t, base.t, self.hooks)
            else:
                old = base[target[2]]
                if isinstance(old, V) and isinstance(value, V) and value.t in ("ai", "af"):
                    value = V(old.t, conv_raw(value.v, value.t, old.t, self.hooks))
                base[target[2]] = value
        else:
            raise TypeError("assignment to %r" % (k,))

    def exec(self, st, scopes):
        k = st[0]
        if k == "block":
            inner = scopes + [{}]
            for s in st[1]:
                yield from self.exec(s, inner)
        elif k in ("let", "var"):
            _, name, ty, init = st
            if init is None:
                val = self.zero(ty)
            else:
                val = self.eval(init, scopes)
                val = self.coerce(val, ty) if ty else self.concrete(val)
                if isinstance(val, (list, dict)):
                    val = copy.deepcopy(val)
                elif isinstance(val, Vec):
                    val = Vec(val.t, val.e)
            scopes[-1][name] = val
        elif k == "assign":
            _, target, op, rhs = st
            if op == "+" and rhs[0] == "bin" and rhs[1] == "*" and self.hooks.contract_muladd:
                # `x += a * b` under a contracting compiler: fma(a, b, x) per component (gaussian_blur_x.wgsl: result += sample * weight)
                x, a, b = self.eval(target, scopes), self.eval(rhs[2], scopes), self.eval(rhs[3], scopes)
                if isinstance(x, Vec) and x.t == "f32":
                    ae = a.e if isinstance(a, Vec) else [conv_raw(a.v, a.t, "f32", self.hooks)] * len(x.e)
                    be = b.e if isinstance(b, Vec) else [conv_raw(b.v, b.t, "f32", self.hooks)] * len(x.e)
                    self.store(target, Vec("f32", [self.hooks.fma(p, q, r) for p, q, r in zip(ae, be, x.e)]), scopes)
                    return
            val = self.eval(rhs, scopes)
            if op is not None:
                val = self.binary(op, self.eval(target, scopes), val)
            self.store(target, val, scopes)
        elif k == "expr":
            e = st[1]
            if e[0] == "call" and e[1] == "workgroupBarrier":
                yield BARRIER
            else:
                self.eval(e, scopes)
        elif k == "if":
            if self.eval(st[1], scopes).v:
                yield from self.exec(st[2], scopes)
            elif st[3] is not None:
                yield from self.exec(st[3], scopes)
        elif k == "for":
            _, init, cond, step, body = st
            inner = scopes + [{}]
            if init is not None:
                yield from self.exec(init, inner)
            while cond is None or self.eval(cond, inner).v:
                yield from self.exec(body, inner)
                if step is not None:
                    yield from self.exec(step, inner)
        elif k == "return":
            raise Return(None if st[1] is None else self.eval(st[1], scopes))
        else:
            raise TypeError("statement %r" % (k,))

    # ---- entry points
    def run_function(self, name, *args):
        """A vertex / fragment entry point (or any function) called once with positional arguments; returns its value."""
        fparams, ret, body, _ = self.funcs[name]
        scope = {p: a for (p, _, _), a in zip(fparams, args)}
        try:
            for _ in self.exec(body, [scope]):
                raise RuntimeError("a barrier outside a compute entry point")
        except Return as r:
            return r.value
        return None

    def dispatch(self, name, groups):
        """A compute entry point over groups = (gx, gy, gz) workgroups; invocations of a workgroup run in local-index order between barriers,
        workgroups one after the other (x fastest) -- one of the orders the API allows."""
        fparams, _, body, attrs = self.funcs[name]
        wx, wy, wz = (int(s) for s in (attrs["workgroup_size"] + ["1", "1"])[:3])
        for gz in range(groups[2]):
            for gy in range(groups[1]):
                for gx in range(groups[0]):
                    self.reset_workgroup()
                    gens = []
                    for lz in range(wz):
                        for ly in range(wy):
                            for lx in range(wx):
                                scope = {}
                                for p, t, a in fparams:
                                    b = a["builtin"][0]
                                    if b == "global_invocation_id":
                                        scope[p] = Vec("u32", [gx * wx + lx, gy * wy + ly, gz * wz + lz])
                                    elif b == "local_invocation_index":
                                        scope[p] = V("u32", lx + wx * (ly + wy * lz))
                                    elif b == "local_invocation_id":
                                        scope[p] = Vec("u32", [lx, ly, lz])
                                    elif b == "workgroup_id":
                                        scope[p] = Vec("u32", [gx, gy, gz])
                                    else:
                                        raise NameError("built-in %s" % b)
                                gens.append(self.exec(body, [scope]))
                    live = gens
                    while live:
                        nxt = []
                        for g in live:
                            try:
                                next(g)
                                nxt.append(g)  # stopped at a barrier
                            except (StopIteration, Return):
                                pass
                        live = nxt
