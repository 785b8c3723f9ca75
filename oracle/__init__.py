"""CPU oracle package — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything from
here; chambers_amd (the product) never does and fails loudly when its HIP library is missing.
See the module headers for parity status (pinned: ImageNetNormalization only)."""
