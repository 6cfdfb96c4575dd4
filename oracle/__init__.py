"""CPU oracle for the Monte Carlo path hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product package (monte_carlo_portfolio_amd) never does.  See mc_oracle.c for the parity status.
"""
