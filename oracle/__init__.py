"""CPU oracle for the multipath-nn hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product path (``multipath-nn_amd/``) never
imports it and fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference delegates all arithmetic to TensorFlow (pre-1.0
API), which is neither installed nor installable here, and the reference ships
no tests, golden vectors or seeds.  What pins this oracle instead:

* known answers derivable from the reference source alone (KA1..KA7,
  SURVEY.md section 8c) -- ``tests/test_oracle_known_answers.py``;
* the one importable reference module (``scripts/lib/data.py``): fixtures
  generated from it by ``tests/golden/make_data_golden.py``;
* an independent implementation: every hand-written float64 NumPy forward and
  backward formula in ``np_ops`` is cross-checked against torch-CPU operators
  and torch autograd (``tests/test_oracle_vs_torch.py``).
"""
