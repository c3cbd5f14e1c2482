"""CPU oracle for the NNToolkitCore time-series inference path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg -- never from ``nntoolkitcore_amd``.
See ``oracle/nnref.h`` for the parity-pinning status of each function.
"""
from .nnref import *  # noqa: F401,F403
