"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (cloops_amd/) never does; it fails loudly without the HIP library.
"""
from .oracle import (build, labels, neighbor_counts, single_dbscan, VARIANTS)  # noqa: F401
