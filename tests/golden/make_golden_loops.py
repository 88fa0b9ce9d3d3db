#!/usr/bin/env python
"""Golden `.loop` table for config 1 (chr21, -m 1): the reference's significance module
(cLoops/cModel.py, converted in memory by tests/refload.py:ref_cmodel_namespace) applied to the 801
candidate records of tests/golden/chr21_pipe.npz, written the way runStat does (pipe.py:191-197).
Build container only:   python tests/golden/make_golden_loops.py"""
import contextlib
import io
import os
import sys
import tempfile

import joblib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refload  # noqa: E402
import golden_util as G  # noqa: E402
import pipe_checks  # noqa: E402


def main():
    ns = refload.ref_cmodel_namespace()
    z, meta = pipe_checks.pipe_golden()
    X, Y = G.chr21_xy()
    mat = np.stack([np.arange(len(X)), X, Y], 1).astype(np.int64)
    for variant in ("v2", "v1"):
        recs = [["chr21", int(a), int(b), "chr21", int(c), int(d)] for a, b, c, d in z[variant + "_filtered"]]
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "chr21-chr21.jd")
            joblib.dump(mat, f)
            with contextlib.redirect_stdout(io.StringIO()):
                ds = ns["getIntSig"](f, recs, [5], 0)          # runStat(dataI, minPts, 0, ...) (pipe.py:284)
            for hic, tag in ((0, ""), (1, "_hic")):
                d2 = (ns["markIntSigHic"] if hic else ns["markIntSig"])(ds.copy())
                out = os.path.join(HERE, "chr21_%s%s.loop" % (variant, tag))
                d2.to_csv(out, sep="\t", index_label="loopId")
                print(variant, tag, d2.shape, int(d2["significant"].sum()))


if __name__ == "__main__":
    main()
