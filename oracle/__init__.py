"""TEST INFRASTRUCTURE ONLY.

CPU restatement (PyTorch fp32 functional ops) of the gandtr hot path, used as the parity checker by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  Nothing under
``gandtr_amd/`` may import this package.
"""
