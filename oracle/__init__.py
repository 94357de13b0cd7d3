"""CPU restatement of the reference algorithm: the parity checker.

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``; never from the product package.
"""
