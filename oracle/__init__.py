"""CPU oracle for the beam hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  It wraps oracle/crb_oracle.c (a plain-C restatement of the reference's
algorithm, pinned by tests/golden/*.npz) through ctypes.
"""
from .oracle import OracleBeam, build, lib  # noqa: F401
