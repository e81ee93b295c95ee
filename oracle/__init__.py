"""CPU oracle for the semtools search hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (semtools_amd/) never does.  See semtools_oracle.h
for the "parity unpinned" statement.
"""
