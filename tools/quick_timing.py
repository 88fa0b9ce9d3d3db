"""Per-kernel HIP-event timings of one clustering run (developer tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import api
from cloops_amd.synth import synth_chrom

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5000000
eps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
minPts = int(sys.argv[3]) if len(sys.argv) > 3 else 5
variant = sys.argv[4] if len(sys.argv) > 4 else "v2"
X, Y = synth_chrom(n, 248956422, 2000)
ch = api.Chromosome(X, Y)
ch.set_profiling(True)
for it in range(4):
    t0 = time.perf_counter()
    res = ch.cluster(variant, eps, minPts)
    t1 = time.perf_counter()
    tm = res.timing
    print("iter %d wall %.2f ms  K=%d | " % (it, (t1 - t0) * 1e3, res.n_clusters) +
          " ".join("%s=%.3f" % (k[3:], v) for k, v in tm.items() if k.startswith("ms_")) +
          " n_in=%d strips=%d" % (tm["n_in"], tm["n_strips"]))
tm = res.timing
res = ch.cluster(variant, eps, minPts, pinned=True); print("pinned:", {k: round(v,3) for k,v in res.timing.items()})
bytes_k2 = tm["n_in"] * 12 + tm["n_strips"] * 4
print("K2: %.1f GB/s algorithmic (%.2f%% of 8 TB/s), %.2f G PETs/s" % (
    bytes_k2 / tm["ms_region"] / 1e6, bytes_k2 / tm["ms_region"] / 1e6 / 8000 * 100, tm["n_in"] / tm["ms_region"] / 1e6))
print("labelled %d" % int((res.labels >= 0).sum()))
