"""CPU oracle for the MPS sweep hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a plain NumPy/SciPy restatement of the
reference's algorithm for the per-site DMRG / TDVP inner loop.  It exists to
*check* the HIP engine (``renormalizer_amd``) and to provide the timed CPU
baseline in ``bench.py``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package never
does, and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the read-only Python
reference (dev container only) and writes input/output vectors captured at the
reference's own seams to ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
replays them against this restatement (<=1e-10) on every CPU test run.
"""
