"""oracle/ — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot-path algorithms.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import anything from
this package; the product (`analysisgnn_amd/`) never does.  See oracle/README.md for what is
pinned against the reference's own code and what is "parity unpinned".
"""
