"""Precondition compiler: boolean expressions over node property names -> postfix program.

The reference evaluates `VulnerabilityInfo.precondition` with the third-party package
boolean.py == 4.0 (src/CyberBattleSim/requirements.txt:5), which is not vendored under
/root/reference; call sites are model.py:219-223 (parse) and actions.py:158-171
(`expr.subs({sym: TRUE|FALSE}).simplify() == TRUE`, a symbol being TRUE iff its name is in
`node.properties`, which also holds the appended `privilege_k` tags).  With every symbol
substituted by a constant the result of simplify() is just the value of the expression, so a
plain evaluator is equivalent.  Restated from boolean.py 4.0's published tokenizer:

    AND  '*' '&' 'and'      OR  '+' '|' 'or'      NOT  '~' '!' 'not'
    '(' ')' '[' ']'         TRUE 'true' '1'       FALSE 'false' '0' 'none'
    keywords are case-insensitive; a symbol starts with a letter or '_' and continues with
    letters, digits, '.', ':' or '_'; binding strength NOT > AND > OR, all left-associative.

This module uses a shunting-yard pass producing a postfix token list.  (The oracle harness's
boolean stand-in is an independently written recursive-descent parser; tests cross-check both.)
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Sequence, Tuple, Union

# postfix tokens: ("sym", name) | ("const", bool) | ("not",) | ("and",) | ("or",)
Token = Tuple

_WORDS = {
    "*": "and", "&": "and", "and": "and",
    "+": "or", "|": "or", "or": "or",
    "~": "not", "!": "not", "not": "not",
    "(": "(", "[": "(", ")": ")", "]": ")",
    "true": True, "1": True,
    "false": False, "0": False, "none": False,
}
_PRIORITY = {"not": 3, "and": 2, "or": 1}


class PreconditionSyntaxError(ValueError):
    pass


def _scan(text: str) -> List[Tuple[str, object]]:
    out: List[Tuple[str, object]] = []
    i, n = 0, len(text)
    while i < n:
        ch = text[i]
        if ch.isalpha() or ch == "_":
            j = i + 1
            while j < n and (text[j].isalnum() or text[j] in "._:"):
                j += 1
            word = text[i:j]
            i = j
            meaning = _WORDS.get(word.lower())
            if meaning is None:
                out.append(("sym", word))
            elif isinstance(meaning, bool):
                out.append(("const", meaning))
            else:
                out.append(("op", meaning))
            continue
        i += 1
        if ch in " \t\r\n":
            continue
        meaning = _WORDS.get(ch)
        if meaning is None:
            raise PreconditionSyntaxError(f"unknown token {ch!r} at position {i - 1} in {text!r}")
        if isinstance(meaning, bool):
            out.append(("const", meaning))
        else:
            out.append(("op", meaning))
    return out


def _to_postfix(tokens: Sequence[Tuple[str, object]], text: str) -> List[Token]:
    output: List[Token] = []
    stack: List[str] = []
    expect_operand = True
    for kind, val in tokens:
        if kind in ("sym", "const"):
            if not expect_operand:
                raise PreconditionSyntaxError(f"missing operator before {val!r} in {text!r}")
            output.append((kind, val))
            expect_operand = False
        elif val == "not":
            if not expect_operand:
                raise PreconditionSyntaxError(f"misplaced NOT in {text!r}")
            stack.append("not")
        elif val == "(":
            if not expect_operand:
                raise PreconditionSyntaxError(f"missing operator before '(' in {text!r}")
            stack.append("(")
        elif val == ")":
            if expect_operand:
                raise PreconditionSyntaxError(f"empty or dangling sub-expression in {text!r}")
            while stack and stack[-1] != "(":
                output.append((stack.pop(),))
            if not stack:
                raise PreconditionSyntaxError(f"unbalanced ')' in {text!r}")
            stack.pop()
        else:  # binary and / or
            if expect_operand:
                raise PreconditionSyntaxError(f"missing operand before {val!r} in {text!r}")
            while stack and stack[-1] != "(" and _PRIORITY[stack[-1]] >= _PRIORITY[val]:
                output.append((stack.pop(),))
            stack.append(val)
            expect_operand = True
    if expect_operand:
        raise PreconditionSyntaxError(f"incomplete expression {text!r}")
    while stack:
        op = stack.pop()
        if op == "(":
            raise PreconditionSyntaxError(f"unbalanced '(' in {text!r}")
        output.append((op,))
    return output


class BoolExpr:
    """A parsed precondition: original text plus its postfix program."""

    __slots__ = ("text", "postfix")

    def __init__(self, text: str, postfix: List[Token]):
        self.text = text
        self.postfix = postfix

    def get_symbols(self) -> List[str]:
        return [t[1] for t in self.postfix if t[0] == "sym"]

    def evaluate(self, is_true: Union[Callable[[str], bool], Iterable[str]]) -> bool:
        if not callable(is_true):
            names = set(is_true)
            is_true = names.__contains__
        stack: List[bool] = []
        for tok in self.postfix:
            k = tok[0]
            if k == "sym":
                stack.append(bool(is_true(tok[1])))
            elif k == "const":
                stack.append(tok[1])
            elif k == "not":
                stack[-1] = not stack[-1]
            else:
                b = stack.pop()
                a = stack.pop()
                stack.append((a and b) if k == "and" else (a or b))
        assert len(stack) == 1
        return stack[0]

    def __str__(self) -> str:
        return self.text

    def __repr__(self) -> str:
        return f"BoolExpr({self.text!r})"


def parse_expression(text: str) -> BoolExpr:
    return BoolExpr(text, _to_postfix(_scan(text), text))


# ---- byte code shared with the CPU oracle (see include/mcbs.h, "precondition byte code") ----
OP_PROP_BASE = 0x00   # 0x00..0x3F  push static property bit i
OP_TAG_BASE = 0x40    # 0x40..0x43  push dynamic tag privilege_k
OP_TRUE = 0x80
OP_FALSE = 0x81
OP_NOT = 0x82
OP_AND = 0x83
OP_OR = 0x84


def encode(expr: BoolExpr, property_index: dict, tag_names: Sequence[str]) -> bytes:
    """Lower to the one-byte-per-op program the oracle interprets.  A symbol that is neither a
    declared property nor a privilege tag can never be in `node.properties` of a validated
    environment, so it lowers to FALSE."""
    code = bytearray()
    for tok in expr.postfix:
        k = tok[0]
        if k == "sym":
            name = tok[1]
            if name in tag_names:
                code.append(OP_TAG_BASE + list(tag_names).index(name))
            elif name in property_index:
                code.append(OP_PROP_BASE + property_index[name])
            else:
                code.append(OP_FALSE)
        elif k == "const":
            code.append(OP_TRUE if tok[1] else OP_FALSE)
        else:
            code.append({"not": OP_NOT, "and": OP_AND, "or": OP_OR}[k])
    return bytes(code)
