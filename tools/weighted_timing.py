import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from cloops_amd import api
from cloops_amd.synth import synth_chrom
X, Y = synth_chrom(5000000, 248956422, 2000)
ch = api.Chromosome(X, Y)
for wx, wy in ((50, 1), (1, 50)):
    for it in range(2):
        t = time.perf_counter(); r = ch.cluster_weighted(20000, 5, wx, wy); dt = time.perf_counter() - t
    print("5M PETs ext (%d,%d): %.2f ms, %d clusters, %d labelled, max scaled coord %.3g" % (wx, wy, dt * 1e3, r.n_clusters, int((r.labels >= 0).sum()), 50.0 * X.max()))
r2 = ch.cluster_weighted(20000, 5, 1, 50)
print("deterministic", bool(np.array_equal(r.labels, r2.labels)))
