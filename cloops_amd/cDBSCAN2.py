"""Drop-in for cLoops/cDBSCAN2.py (reference class `cDBSCAN`, cDBSCAN2.py:7-383) on MI355X.

This is the production variant (`from cLoops.cDBSCAN2 import cDBSCAN as DBSCAN`,
cLoops/pipe.py:42).  Needs 0 <= X <= Y (guaranteed by cLoops/io.py:49-57)."""
from ._dbscan_base import _GpuDBSCAN


class cDBSCAN(_GpuDBSCAN):
    _variant = "v2"
