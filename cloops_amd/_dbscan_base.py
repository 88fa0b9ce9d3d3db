"""Shared implementation of the three reference-shaped clustering classes."""
import numpy as np

from . import api


class _GpuDBSCAN(object):
    """`cls(mat, eps, minPts)` does all the work; `.labels` is the reference's result dict
    {pointId: clusterId} holding clustered points only.  The dict is materialised lazily
    from `.labels_array` (int32, aligned to mat's rows, -1 = absent), because building a
    dict of millions of entries costs more than the clustering itself."""

    _variant = None

    def __init__(self, mat, eps, minPts, device=0):
        self.eps = eps
        self.minPts = minPts
        self.cw = self.eps                      # cDBSCAN.py:29 / cDBSCAN2.py:30
        mat = np.asarray(mat)
        if mat.ndim != 2 and len(mat) > 0:
            raise IndexError("mat must be [N,3] rows of [pointId, X, Y]")
        n = len(mat)
        self._labels = None
        if n == 0:
            if self._variant != "v2":
                # cDBSCAN.py:77 / blockDBSCAN.py:74: `mat[0]` on an empty mat
                raise IndexError("index 0 is out of bounds for axis 0 with size 0")
            self.ids = np.zeros(0, np.int64)
            self.labels_array = np.zeros(0, np.int32)
            self.boxes = np.zeros(0, api.BOX_DTYPE)
            return
        if int(eps) != eps:
            raise TypeError("eps must be an integer (all reference presets are, pipe.py:329-344)")
        if eps == 0:
            raise ZeroDivisionError("division by zero")       # int(x / self.cw)
        self.ids = mat[:, 0]
        chrom = api.Chromosome(mat[:, 1], mat[:, 2], device=device)
        try:
            res = chrom.cluster(self._variant, int(eps), int(minPts), 0)
        finally:
            chrom.close()
        self.labels_array = res.labels
        self.boxes = res.boxes
        self.n_clusters = res.n_clusters

    @property
    def labels(self):
        if self._labels is None:
            lab = self.labels_array
            sel = np.nonzero(lab >= 0)[0]
            ids = self.ids[sel]
            self._labels = dict(zip(ids.tolist() if ids.dtype.kind != "i" else ids, lab[sel].tolist()))
        return self._labels

    @labels.setter
    def labels(self, value):
        self._labels = value
