#!/usr/bin/env python
"""Golden labels for the stripe clustering of scripts/callStripes:37-52: the REAL reference class
cLoops.cDBSCAN (variant 1) on the chr21 example matrix with the X (resp. Y) column multiplied by
ext = 50, eps = 20000, minPts = 5 (the script's defaults, scripts/callStripes:281-286).
Build container only:   python tests/golden/make_golden_stripes.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refload  # noqa: E402
import golden_util as G  # noqa: E402


def main():
    X, Y = G.chr21_xy()
    ids = np.arange(len(X), dtype=np.int64)
    out = {}
    for name, wx, wy in (("x50", 50, 1), ("y50", 1, 50)):
        mat = np.stack([ids, X.astype(np.int64) * wx, Y.astype(np.int64) * wy], 1)
        lab = refload.labels_dict_to_array(refload.ref_labels("v1", mat, 20000, 5), ids)
        out[name] = lab
        print(name, "clusters", len(np.unique(lab[lab >= 0])), "labelled", int((lab >= 0).sum()), "max scaled coordinate", int(mat[:, 1:].max()))
    np.savez_compressed(os.path.join(HERE, "chr21_stripes_labels.npz"), **out)


if __name__ == "__main__":
    main()
