"""Stand-in for the third-party package boolean.py (pinned ==4.0 by the reference:
src/CyberBattleSim/requirements.txt:5, requirements.txt:11; absent here, no network).

TEST TOOLING ONLY.  Restates the published boolean.py 4.0 behaviour for the surface the
reference's hot path uses (model.py:33,44,219-223; actions.py:14,111,165-171):
BooleanAlgebra.parse, Expression.get_symbols / subs / simplify / __eq__ / __str__.
Deliberately written as a recursive-descent parser, independent of the product's
shunting-yard compiler in marlon_amd/precondition.py, so the two can be cross-checked.
"""
from . import boolean  # noqa: F401
from .boolean import BooleanAlgebra, Expression, Symbol, ParseError  # noqa: F401
