"""CPU oracle for the SwinGNN denoiser hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / reported baseline.
The product path (``swingnn_amd``) never imports this package and raises if the
HIP library is missing.

Parity status: PINNED.  ``oracle/swingnn_ref.py`` is checked against golden
vectors produced by importing the reference's own modules in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
``tests/test_oracle_golden.py``).
"""
