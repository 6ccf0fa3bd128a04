"""CPU oracle for the KING hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / timed CPU baseline.
The product (``cuking_amd``) never does.

PARITY UNPINNED: the reference ships no golden vectors and cannot be built or
imported here; see ``king_oracle.h`` for what pins this restatement instead.

Contents
--------
``king_oracle.c``   C restatement of cuking.cu's arithmetic (cited per function).
``synth_oracle.c``  CPU twin of the synthetic-input generator.
``naive_oracle.py`` per-genotype numpy oracle, no bitsets (independent check).
``pyoracle.py``     ctypes bindings for the two C files.
"""
