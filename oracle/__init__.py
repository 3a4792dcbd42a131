"""TEST INFRASTRUCTURE -- the CPU checker for the HIP engine (see oracle/vpic_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
