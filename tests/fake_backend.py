"""CPU stand-in for cloops_amd.api.Chromosome used ONLY by the `not gpu` tests of the host
logic (pipe.py, wrappers): same interface, results from the CPU oracle.  Installed by
monkeypatching in the tests; the product never sees it."""
import numpy as np

import oracle
from cloops_amd import api


class FakeChromosome(object):
    def __init__(self, X, Y, device=0, stream=None):
        self.X = np.asarray(X, np.int64)
        self.Y = np.asarray(Y, np.int64)
        self.n = len(self.X)
        self.device = device

    def close(self):
        pass

    def set_profiling(self, on=True):
        pass

    def cluster(self, variant, eps, minPts, cut=0, want_labels=True, want_boxes=True, pinned=False):
        vname = {1: "v1", 2: "v2", 3: "block"}[api.VARIANTS[variant]]
        lab = oracle.single_dbscan(vname, self.X, self.Y, eps, minPts, cut)["labels"]
        ml = int(lab.max()) if len(lab) else -1
        boxes = np.zeros(ml + 1, api.BOX_DTYPE)
        for c in np.unique(lab[lab >= 0]):
            m = lab == c
            boxes[c] = (self.X[m].min(), self.X[m].max(), self.Y[m].min(), self.Y[m].max(), m.sum())
        return api.ClusterResult(lab, int(len(np.unique(lab[lab >= 0]))), ml, boxes, None)
