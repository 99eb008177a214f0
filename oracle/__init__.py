"""CPU oracle for the CSR x vector / Lanczos / CG hot path -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under quantum_basis_amd/ does.
"""
