"""Drop-in for cLoops/blockDBSCAN.py (reference class `blockDBSCAN`, blockDBSCAN.py:6-239)
on MI355X (the alternative import at cLoops/pipe.py:43)."""
from ._dbscan_base import _GpuDBSCAN


class blockDBSCAN(_GpuDBSCAN):
    _variant = "block"
