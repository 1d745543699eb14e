"""CPU oracle for the CLIP training hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the reported CPU baseline -- never as the thing that
is measured or shipped.  The product path (``open_clip_amd``) fails loudly when
the HIP extension is missing; it never falls back to this code.
"""
