"""CPU oracle for the NRMS hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in `news_recommendation_amd/` (the product) may import this package.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
use it, and only as the checker / the reported CPU baseline.

Parity status: PINNED.  The restatements below are checked against golden
vectors produced by importing the reference's own modules from
/root/reference/src in the build container (see oracle/make_golden.py; the
vectors live in tests/golden/, the checks in tests/test_oracle_golden.py).
The reference itself ships no tests or known-answer vectors.
"""
