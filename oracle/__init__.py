"""CPU oracle for the LINEMOD matching path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See oracle/linemod_oracle.cpp for what it restates and why parity is unpinned.
"""
