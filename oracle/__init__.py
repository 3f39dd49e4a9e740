"""CPU oracle of the reference's step semantics — TEST INFRASTRUCTURE, not part of the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything from here.
Parity status: PINNED — against the reference's own known answers and against golden traces captured
from the imported reference (see oracle/cbs_oracle.c header and DESIGN.md section "Oracle").
"""
