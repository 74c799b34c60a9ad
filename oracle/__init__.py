"""CPU oracle for the newmap `search` hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``newmap_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, and only as
the checker / the timed CPU baseline -- never as the thing shipped.
"""
