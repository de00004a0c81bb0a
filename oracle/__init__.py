"""CPU oracle for the IF-Net occupancy-query hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker / the timed CPU baseline.  The shipped
path (``single-view-3d-reconstruction_amd``) never imports this package and
fails loudly when its HIP library is missing.

Parity pinning: the reference has no tests and no golden vectors of its own
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
imported and run in the build container by ``oracle/gen_golden.py``; the vectors
live in ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks the
oracle against them on every CPU run.
"""
