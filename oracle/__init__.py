"""CPU oracle for the yacrd bad-region path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  See oracle/yacrd_oracle.h for the parity pin.
"""
from .oracle import (  # noqa: F401
    NOT_BAD, CHIMERIC, NOT_COVERED, TYPE_NAMES,
    build, compute_bad_part, type_of_read, run, parse_paf, parse_m4, to_csr,
    report_line, report_lines, report_from_csr,
)
