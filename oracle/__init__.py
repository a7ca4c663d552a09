"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path (``libclap_oracle.so``, built from
the C files in this directory) and a runner for the real reference code
(``oracle/_ref/clap_ref``, built by ``oracle/ref/Makefile`` in the build
container).  Imported only by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg -- never by ``clap_amd``.
"""
