"""CPU oracle of the sketch + pairwise hot path.  TEST INFRASTRUCTURE ONLY: may be imported from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the product package."""
