"""CPU oracle for the Vlaser hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A dependency-light PyTorch-CPU restatement (written from the math, not copied) of the reference functions
listed in SURVEY.md §8a.  It is the checker the HIP path is compared against, and the `cpu_baseline` leg of
bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it; the product
package `vlaser_amd` never does (tests/test_layout.py enforces that).

Pinning: the reference ships no tests or golden vectors for this path, so the oracle is
pinned against outputs of the reference itself, produced in the build container by tools/gen_golden.py
(which imports /root/reference with the stubs of tools/ref_import.py) and committed under tests/golden/.
tests/test_oracle_golden.py replays them.
"""
