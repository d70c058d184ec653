"""CPU oracle for the LectureMath hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (lecturemath_amd) never does; it fails loudly when its HIP library is missing.

Parity status: pinned against outputs of the reference itself run in the build container
(tests/golden/make_golden.py; tests/test_oracle_golden.py, tests/test_reference_binding.py).
"""
