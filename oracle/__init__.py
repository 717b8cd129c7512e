"""CPU oracle package — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
parl_amd/ never does (tests/test_capi_symbols.py::test_no_oracle_in_product enforces it)."""
