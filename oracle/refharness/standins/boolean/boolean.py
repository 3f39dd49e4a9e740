"""boolean.py 4.0 restatement (subset).  Grammar and token table follow the published
tokenizer: AND = '*' '&' 'and'; OR = '+' '|' 'or'; NOT = '~' '!' 'not'; parentheses
'(' ')' '[' ']'; TRUE = 'true' '1'; FALSE = 'false' '0' 'none' (keywords case-insensitive);
a symbol starts with a letter or '_' and continues with alphanumerics, '.', ':' or '_'.
Binding strength: NOT > AND > OR."""


class ParseError(Exception):
    pass


class Expression:
    def get_symbols(self):
        out = []
        self._collect(out)
        return out

    @property
    def symbols(self):
        return set(self.get_symbols())

    def subs(self, mapping, default=None, simplify=False):
        e = self._subs(mapping)
        return e.simplify() if simplify else e

    def simplify(self):
        return self

    def __ne__(self, other):
        return not self.__eq__(other)


class _Const(Expression):
    def __init__(self, value):
        self.value = bool(value)

    def _collect(self, out):
        pass

    def _subs(self, mapping):
        return self

    def __eq__(self, other):
        return isinstance(other, _Const) and other.value == self.value

    def __hash__(self):
        return hash(("const", self.value))

    def __str__(self):
        return "1" if self.value else "0"

    __repr__ = __str__


TRUE = _Const(True)
FALSE = _Const(False)


class Symbol(Expression):
    def __init__(self, obj):
        self.obj = obj

    def _collect(self, out):
        out.append(self)

    def _subs(self, mapping):
        return mapping.get(self, self)

    def __eq__(self, other):
        return isinstance(other, Symbol) and other.obj == self.obj

    def __hash__(self):
        return hash(("sym", self.obj))

    def __str__(self):
        return str(self.obj)

    def __repr__(self):
        return f"Symbol({self.obj!r})"


class NOT(Expression):
    def __init__(self, arg):
        self.args = (arg,)

    def _collect(self, out):
        self.args[0]._collect(out)

    def _subs(self, mapping):
        return NOT(self.args[0]._subs(mapping))

    def simplify(self):
        a = self.args[0].simplify()
        if isinstance(a, _Const):
            return FALSE if a.value else TRUE
        if isinstance(a, NOT):
            return a.args[0]
        return NOT(a)

    def __eq__(self, other):
        return isinstance(other, NOT) and other.args == self.args

    def __hash__(self):
        return hash(("not", self.args))

    def __str__(self):
        a = self.args[0]
        return f"~{a}" if isinstance(a, (Symbol, _Const)) else f"~({a})"


class _Dual(Expression):
    op = "?"
    identity = TRUE
    annihilator = FALSE

    def __init__(self, *args):
        self.args = tuple(args)

    def _collect(self, out):
        for a in self.args:
            a._collect(out)

    def _subs(self, mapping):
        return type(self)(*[a._subs(mapping) for a in self.args])

    def simplify(self):
        kept = []
        for a in self.args:
            a = a.simplify()
            if a == self.annihilator:
                return self.annihilator
            if a == self.identity:
                continue
            if isinstance(a, type(self)):
                kept.extend(x for x in a.args if x not in kept)
            elif a not in kept:
                kept.append(a)
        for a in kept:
            if NOT(a).simplify() in kept:
                return self.annihilator
        if not kept:
            return self.identity
        if len(kept) == 1:
            return kept[0]
        return type(self)(*kept)

    def __eq__(self, other):
        return type(other) is type(self) and set(other.args) == set(self.args)

    def __hash__(self):
        return hash((self.op, frozenset(self.args)))

    def __str__(self):
        parts = []
        for a in self.args:
            s = str(a)
            parts.append(s if isinstance(a, (Symbol, _Const, NOT)) else f"({s})")
        return self.op.join(parts)


class AND(_Dual):
    op = "&"
    identity = TRUE
    annihilator = FALSE


class OR(_Dual):
    op = "|"
    identity = FALSE
    annihilator = TRUE


_KEYWORDS = {
    "*": "AND", "&": "AND", "and": "AND",
    "+": "OR", "|": "OR", "or": "OR",
    "~": "NOT", "!": "NOT", "not": "NOT",
    "(": "LPAR", ")": "RPAR", "[": "LPAR", "]": "RPAR",
    "true": "TRUE", "1": "TRUE",
    "false": "FALSE", "0": "FALSE", "none": "FALSE",
}


def _tokenize(text):
    pos, n = 0, len(text)
    while pos < n:
        tok = text[pos]
        is_sym = tok.isalpha() or tok == "_"
        if is_sym:
            pos += 1
            while pos < n and (text[pos].isalnum() or text[pos] in ".:_"):
                tok += text[pos]
                pos += 1
            pos -= 1
        kind = _KEYWORDS.get(tok.lower())
        if kind is not None:
            yield kind, tok
        elif is_sym:
            yield "SYMBOL", tok
        elif tok not in " \t\r\n":
            raise ParseError(f"unknown token {tok!r} at {pos}")
        pos += 1


class BooleanAlgebra:
    def __init__(self):
        self.TRUE = TRUE
        self.FALSE = FALSE
        self.Symbol = Symbol
        self.NOT, self.AND, self.OR = NOT, AND, OR

    def parse(self, expr, simplify=False):
        if isinstance(expr, Expression):
            return expr
        toks = list(_tokenize(expr))
        self._t, self._i = toks, 0
        if not toks:
            raise ParseError("empty expression")
        e = self._or()
        if self._i != len(toks):
            raise ParseError(f"unexpected token {toks[self._i][1]!r}")
        return e.simplify() if simplify else e

    def _peek(self):
        return self._t[self._i][0] if self._i < len(self._t) else None

    def _or(self):
        args = [self._and()]
        while self._peek() == "OR":
            self._i += 1
            args.append(self._and())
        return args[0] if len(args) == 1 else OR(*args)

    def _and(self):
        args = [self._not()]
        while self._peek() == "AND":
            self._i += 1
            args.append(self._not())
        return args[0] if len(args) == 1 else AND(*args)

    def _not(self):
        if self._peek() == "NOT":
            self._i += 1
            return NOT(self._not())
        return self._atom()

    def _atom(self):
        k = self._peek()
        if k is None:
            raise ParseError("unexpected end of expression")
        tok = self._t[self._i][1]
        self._i += 1
        if k == "LPAR":
            e = self._or()
            if self._peek() != "RPAR":
                raise ParseError("missing closing parenthesis")
            self._i += 1
            return e
        if k == "TRUE":
            return TRUE
        if k == "FALSE":
            return FALSE
        if k == "SYMBOL":
            return Symbol(tok)
        raise ParseError(f"unexpected token {tok!r}")
