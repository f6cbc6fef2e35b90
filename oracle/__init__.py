"""CPU oracle for the PDDP/iLQR hot path - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (pddp_amd) never does.  See pddp_oracle.h.
"""
from .oracle import (Oracle, load, make_problem, PddpProblem, PROBLEM_NAMES,
                     build)

__all__ = ["Oracle", "load", "make_problem", "PddpProblem", "PROBLEM_NAMES",
           "build"]
