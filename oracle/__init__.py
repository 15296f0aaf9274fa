"""CPU oracle for the frame-synthesis hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  See vfi_oracle.h for the parity status
("parity unpinned" by the reference; pinned by analytic cases + np_oracle).
"""
