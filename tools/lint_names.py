#!/usr/bin/env python3
"""Names used but never bound (the typo class a GPU-less container cannot hit at run time): for every function scope of the given
files, a name the compiler resolves as a global must be bound at module level or be a builtin.

    python tools/lint_names.py [files...]        exit code 1 when something is unbound (tests/test_host_logic.py runs it)"""
import builtins
import os
import symtable
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["bench.py", "__graft_entry__.py"] + [os.path.join(d, f) for d in ("tools", "marlon_amd", os.path.join("marlon_amd", "samples"), "oracle")
                                                 for f in sorted(os.listdir(os.path.join(REPO, d))) if f.endswith(".py")]


def unbound(path: str):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    module_names = set(top.get_identifiers())
    out = []

    def walk(t):
        for c in t.get_children():
            for s in c.get_symbols():
                n = s.get_name()
                if s.is_global() and s.is_referenced() and n not in module_names and not hasattr(builtins, n):
                    out.append((c.get_name(), n))
            walk(c)
    walk(top)
    return out


def main(files):
    bad = 0
    for f in files:
        for scope, name in unbound(f if os.path.isabs(f) else os.path.join(REPO, f)):
            print(f"{f}: in {scope}: name '{name}' is never bound")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or DEFAULT))
