"""Imported by bench.py through CLOOPS_BENCH_PRELOAD in the CPU tests only: replaces the GPU handle by the
oracle-backed stand-in so that the launch / sharding / exchange logic of bench.py can run without a GPU."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
for p in (_HERE, os.path.dirname(_HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

import fake_backend                      # noqa: E402
from cloops_amd import api               # noqa: E402

api.Chromosome = fake_backend.FakeChromosome
api.device_count = lambda: 1
