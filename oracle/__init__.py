"""CPU oracle for the moment-matched GP rollout (TEST INFRASTRUCTURE ONLY).

Nothing under ``oracle/`` is part of the product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.
"""
