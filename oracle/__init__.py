"""ORACLE — test infrastructure only.

CPU restatements of the reference's algorithms for the VIST3A text->3DGS inference path.  Nothing on the
product path (vist3a_amd/, inference_t23d.py) may import this package; only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do, and only as the checker.  Each module cites the reference file:line it
restates and states whether its parity is pinned by golden vectors generated from the reference itself
(tests/golden/, generator: tests/golden/make_golden.py) or unpinned (DiT / scheduler: third-party
diffusers==0.33.1, absent from /root/reference)."""
