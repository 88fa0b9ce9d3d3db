"""Drop-in for cLoops/cDBSCAN.py (reference class `cDBSCAN`, cDBSCAN.py:6-205) on MI355X.

Same constructor `cDBSCAN(mat, eps, minPts)` and same `.labels` result; the clustering
itself is the HIP library (variant 1).  Callers in the reference: scripts/callStripes:29,51,
scripts/jd2saturation:23,69."""
from ._dbscan_base import _GpuDBSCAN


class cDBSCAN(_GpuDBSCAN):
    _variant = "v1"
