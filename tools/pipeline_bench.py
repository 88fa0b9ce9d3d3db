"""Stage timings of the whole flow (sweep -> significance -> .loop) on a synthetic genome, one GPU.

    python tools/pipeline_bench.py [n_total] [mode]
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import pipe, cModel
from cloops_amd.synth import synth_genome

n_total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eps, minPts, hic = pipe.MODES[mode]
fs = []
for name, X, Y in synth_genome(n_total, cfg=mode):
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
pipe.runSweepFast(fs[:1], eps[:1], minPts[:1], cut=0)          # warm-up (allocations)
t0 = time.perf_counter()
dataI, cut, cuts, steps = pipe.runSweepFast(fs, eps, minPts, cut=0)
t1 = time.perf_counter()
records = {key: {"f": v["f"], "records": pipe._records(key, v["boxes"])} for key, v in dataI.items()}
t2 = time.perf_counter()
with tempfile.TemporaryDirectory() as td:
    fout = os.path.join(td, "out")
    e = cModel.runStat(records, minPts, 0, 1, fout, hic)
    t3 = time.perf_counter()
    nrows = sum(1 for _ in open(fout + ".loop")) - 1 if e == 0 else 0
ncand = sum(len(v["records"]) for v in records.values())
print("PETs %d, mode %d: sweep %.2f s (%d steps, final cut %d) | records %.2f s (%d candidates) | significance + .loop %.2f s (%d rows)" % (
    n_total, mode, t1 - t0, len(steps), cut, t2 - t1, ncand, t3 - t2, nrows))
