"""CPU oracle for the Cattus self-play hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in ``cattus_amd`` may import this package.  It is loaded by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` as the
checker / reported CPU baseline, never as a product code path.
"""
