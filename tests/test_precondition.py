"""Precondition compiler (marlon_amd/precondition.py, shunting-yard) against the independently written
recursive-descent restatement of boolean.py 4.0 used by the oracle harness, on the reference's own expressions
and on random ones; plus the byte code the C oracle interprets."""
import itertools
import os
import sys

import pytest
from hypothesis import given, settings, strategies as st

from marlon_amd import precondition as pc

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle", "refharness", "standins"))
from boolean import boolean as standin  # noqa: E402

ALG = standin.BooleanAlgebra()

REFERENCE_EXPRESSIONS = [
    "true", "SasUrlInCommit&GitHub",                                                 # model.py:237, toy_ctf.py:108
    "Windows&Win10&(~(privilege_2|privilege_3))", "Windows&(privilege_2|privilege_3)", "Windows&PortRDPOpen",  # actions_test.py:39-73
    "Windows&(Win10|Win7)&(~(privilege_2|privilege_3))",                             # environment_generation_test.py:38-44
    "A&B|~C", "not A or B and C", "A * B + !C", "[A | B] & C", "FALSE | a.b:c_d", "1&0|1", "~~A", "none|A",
]


def standin_value(text, true_names):
    e = ALG.parse(text)
    mapping = {s: (standin.TRUE if str(s) in true_names else standin.FALSE) for s in e.get_symbols()}
    return e.subs(mapping).simplify() == standin.TRUE     # the expression of actions.py:169-171


@pytest.mark.parametrize("text", REFERENCE_EXPRESSIONS)
def test_same_truth_table_as_boolean_py_restatement(text):
    e = pc.parse_expression(text)
    syms = sorted(set(e.get_symbols()))
    assert syms == sorted({str(s) for s in ALG.parse(text).get_symbols()})
    for bits in itertools.product([0, 1], repeat=len(syms)):
        names = {s for s, b in zip(syms, bits) if b}
        assert e.evaluate(names) == standin_value(text, names), (text, names)


NAMES = ["A", "B", "C", "Win10", "privilege_2"]
leaf = st.sampled_from(NAMES + ["true", "false", "1", "0"])
expr = st.recursive(leaf, lambda kids: st.one_of(
    st.tuples(st.sampled_from(["~", "!", "not "]), kids).map(lambda t: f"{t[0]}({t[1]})"),
    st.tuples(kids, st.sampled_from(["&", "|", " and ", " or ", "*", "+"]), kids).map(lambda t: f"{t[0]}{t[1]}{t[2]}"),
    st.tuples(kids, st.sampled_from(["&", "|"]), kids).map(lambda t: f"({t[0]}{t[1]}{t[2]})")), max_leaves=8)


@settings(max_examples=300, deadline=None)
@given(expr, st.sets(st.sampled_from(NAMES)))
def test_random_expressions_agree(text, names):
    assert pc.parse_expression(text).evaluate(names) == standin_value(text, names)


def test_precedence_and_errors():
    assert pc.parse_expression("A|B&C").evaluate({"A"}) is True          # AND binds tighter than OR
    assert pc.parse_expression("~A&B").evaluate({"B"}) is True           # NOT binds tightest
    for bad in ["", "A&", "&A", "(A", "A)", "A B", "A~B", "A $ B"]:
        with pytest.raises(ValueError):
            pc.parse_expression(bad)


def test_byte_code():
    e = pc.parse_expression("Windows&~privilege_2|Ghost")
    code = pc.encode(e, {"Windows": 3}, tuple(f"privilege_{k}" for k in range(4)))
    assert list(code) == [pc.OP_PROP_BASE + 3, pc.OP_TAG_BASE + 2, pc.OP_NOT, pc.OP_AND, pc.OP_FALSE, pc.OP_OR]
