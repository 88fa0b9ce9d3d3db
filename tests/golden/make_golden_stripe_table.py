#!/usr/bin/env python
"""Golden `.stripe` tables for the chr21 example: the reference's own stripe functions
(scripts/callStripes, converted in memory by tests/refload.py:ref_stripes_namespace) run the way
callStripes() chains them (scripts/callStripes:331-372): singleStripDBSCAN (ext on X, then on Y) ->
filterCandidateStripes -> estStripeSig -> markStripeSig -> to_csv.  eps 20000, minPts 5, ext 50 are the
script's defaults; pets / lengthFoldDiff are lowered (60 / 8) because the example is one small chromosome.
Build container only:   python tests/golden/make_golden_stripe_table.py"""
import contextlib
import io
import json
import os
import sys
import tempfile

import joblib
import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refload  # noqa: E402
import golden_util as G  # noqa: E402

EPS, MINPTS, EXT, PETS, LENFOLD = 20000, 5, 50, 60, 8


def main():
    ns = refload.ref_stripes_namespace()
    X, Y = G.chr21_xy()
    mat = np.stack([np.arange(len(X)), X, Y], 1).astype(np.int64)
    meta = {"eps": EPS, "minPts": MINPTS, "ext": EXT, "pets": PETS, "lengthFoldDiff": LENFOLD}
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "chr21-chr21.jd")
        joblib.dump(mat, f)
        for name, kw in (("x_horizontal", {"extx": EXT}), ("y_vertical", {"exty": EXT})):
            with contextlib.redirect_stdout(io.StringIO()):
                key, dataI = ns["singleStripDBSCAN"](f, EPS, MINPTS, **kw)
                ds = ns["filterCandidateStripes"]({key: dataI}, pets=PETS, lengthFoldDiff=LENFOLD)
                tab = ns["estStripeSig"](f, ds[key])
            meta[name + "_clusters"] = len(dataI)
            meta[name + "_candidates"] = len(ds[key])
            tab = ns["markStripeSig"](pd.concat([tab]))
            out = os.path.join(HERE, "chr21_%s.stripe" % name)
            tab.to_csv(out, sep="\t", index_label="stripeId")
            meta[name + "_significant"] = int(tab["significant"].sum())
            print(name, meta[name + "_clusters"], "clusters,", meta[name + "_candidates"], "candidates,", meta[name + "_significant"], "significant")
    with open(os.path.join(HERE, "chr21_stripes_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
