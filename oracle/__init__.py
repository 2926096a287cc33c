"""ORACLE — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package; ``dvis_plus_amd`` (the product) never does.
Parity status: pinned — every function is checked in ``tests/test_oracle.py`` against
golden vectors generated from the imported reference (``tests/golden/gen_golden.py``).
"""
