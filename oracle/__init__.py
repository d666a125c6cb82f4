"""CPU oracle for the cDDPM sampling hot path — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy float64 for the schedule/index
arithmetic, plain torch-CPU fp32 ops for the network) of the reference
algorithm on the path named by BASELINE.json's north_star.  Every function
cites the reference file:line it follows.

Rules (see DESIGN.md "Oracle"):
  * Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import it, and only as the checker / the timed CPU baseline.
  * The product package (``diffusion_models_dsdiff_amd``) never imports it and
    has no CPU fallback: it raises when the HIP library is missing.
  * Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing
    the reference itself on CPU in the build container
    (``tests/golden/gen_golden.py``); ``tests/test_oracle_golden.py`` checks the oracle
    against every one of them (integer maps bit-exact, tables to float64/fp32
    equality, network outputs to <= 2e-6 rel-L2).
"""
