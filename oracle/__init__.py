"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/dmpc_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
