"""CPU oracle: TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference algorithm for the hot path, written from the
reference's behaviour (each function cites the reference file:line it follows) and pinned against
golden vectors produced by importing the reference itself in the build container
(tests/golden/gen_golden.py; fixtures under tests/golden/*.npz).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under cet_pick_amd/ imports it; the product path fails loudly when the HIP library is missing.
"""
