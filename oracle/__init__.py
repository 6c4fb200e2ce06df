"""CPU oracle for the Longbow k-NN hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (longbow_amd/) never does.
"""
