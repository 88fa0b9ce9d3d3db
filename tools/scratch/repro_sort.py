import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cloops_amd import api
k = 120000
cx = np.arange(k, dtype=np.int64) * 1000 + 5000
j = np.arange(3, dtype=np.int64)
X = np.concatenate([(cx[:, None] + j[None, :]).ravel(), (np.arange(2000, dtype=np.int64)[:, None] * 700 + 100 + j[None, :]).ravel()])
Y = np.concatenate([(cx[:, None] + 50000 + 2 * j[None, :]).ravel(), (np.arange(2000, dtype=np.int64)[:, None] * 700 + 101 + j[None, :]).ravel()])
ch = api.Chromosome(X.astype(np.int32), Y.astype(np.int32))
for eps in (10, 20):
    r = ch.cluster("v2", eps, 3, 0)
    print("eps", eps, "clusters", r.n_clusters, flush=True)
