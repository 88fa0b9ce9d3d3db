"""One chromosome of the 200 M-PET synthetic genome (BASELINE.json configs[3]) at one (eps, minPts, cut):
the regime the metric's target is written for.  Run under rocprofv3 by tools/profile_dense.sh.

    python tools/dense_run.py [eps] [minPts] [cut] [reps] [chrom_index] [n_total]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import api
from cloops_amd.synth import synth_chrom, chrom_sizes

eps = int(sys.argv[1]) if len(sys.argv) > 1 else 7500
minPts = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cut = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ci = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n_total = int(float(sys.argv[6])) if len(sys.argv) > 6 else 200000000
NOLABELS = os.environ.get("DENSE_NOLABELS") == "1"       # the sweep's form: no row-aligned labels, table only
name, length, n = chrom_sizes(n_total)[ci]
X, Y = synth_chrom(n, length, 1000 * 3 + ci)          # the cfg-3 genome of cloops_amd.synth.synth_genome
ch = api.Chromosome(X, Y)
ch.set_profiling(True)
if NOLABELS:
    ch.set_device_labels(False)
for it in range(reps):
    t0 = time.perf_counter()
    res = ch.cluster("v2", eps, minPts, cut, pinned=True, want_labels=not NOLABELS)
    t1 = time.perf_counter()
    tm = res.timing
    print("%s n=%d eps=%d minPts=%d cut=%d iter %d wall %.2f ms K=%d | " % (name, n, eps, minPts, cut, it, (t1 - t0) * 1e3, res.n_clusters) +
          " ".join("%s=%.3f" % (k[3:], v) for k, v in tm.items() if k.startswith("ms_")) +
          " n_in=%d strips=%d" % (tm["n_in"], tm["n_strips"]))
b = tm["n_in"] * 12 + tm["n_strips"] * 4
k2 = max(tm["ms_region"] - tm["ms_bracket"], 1e-6)
print("K2: %.3f ms -> %.1f GB/s algorithmic (%.2f %% of 8 TB/s)" % (k2, b / k2 / 1e6, b / k2 / 1e6 / 80))
