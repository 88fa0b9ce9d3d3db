#!/usr/bin/env python
"""Golden vectors for the dispatch layer (cLoops/pipe.py:52-174, 241-281), produced by the
reference's OWN functions (extracted in memory by tests/refload.py:ref_pipe_namespace) on
the chr21 example -- run in the build container only:

    python tests/golden/make_golden_pipe.py
"""
import contextlib
import io
import json
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


def boxes(records):
    return np.asarray([[r[1], r[2], r[4], r[5]] for r in records], dtype=np.int64).reshape(-1, 4)


def main():
    X, Y = G.chr21_xy()
    mat = np.stack([np.arange(len(X)), X, Y], 1).astype(np.int64)
    out = {}
    meta = {}
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "chr21-chr21.jd")
        joblib.dump(mat, f)
        for variant in ("v2", "v1"):
            ns = refload.ref_pipe_namespace(variant)
            # the sweep of pipe.py:241-281 for mode 1 (eps 500,1000,2000; minPts 5), cut chained
            dataI, cuts, cut = {}, [0], 0
            steps = []
            for ep in (500, 1000, 2000):
                for m in (5,):
                    err = io.StringIO()
                    with contextlib.redirect_stderr(err):
                        dataI_2, dataS_2, dis_2, dss_2 = ns["runDBSCAN"]([f], ep, m, cut, 1)
                    tag = "%s_%d_%d" % (variant, ep, m)
                    recs = dataI_2[("chr21", "chr21")]["records"] if dataI_2 else []
                    out[tag + "_dataI"] = boxes(recs)
                    out[tag + "_dataS"] = boxes(dataS_2)
                    out[tag + "_dis_sorted"] = np.sort(np.asarray(dis_2, dtype=np.int64))
                    out[tag + "_dss_sorted"] = np.sort(np.asarray(dss_2, dtype=np.int64))
                    st = {"eps": ep, "minPts": m, "cut_in": int(cut), "stderr": err.getvalue()}
                    if len(dataI_2) == 0:
                        steps.append(st)
                        continue
                    if len(dis_2) == 0 or len(dss_2) == 0:
                        dataI = ns["combineTwice"](dataI, dataI_2)
                    else:
                        cut_2, frags = ns["estIntSelCutFrag"](np.array(dis_2), np.array(dss_2))
                        st["cut_out"] = int(cut_2)
                        st["frags"] = int(frags)
                        cuts.append(cut_2)
                        cut = cut_2
                        dataI = ns["combineTwice"](dataI, dataI_2)
                    steps.append(st)
            cuts = [c for c in cuts if c > 0]
            final_cut = int(np.min(cuts))
            n_before = len(dataI[("chr21", "chr21")]["records"])
            out[variant + "_combined"] = boxes(dataI[("chr21", "chr21")]["records"])
            dataI = ns["filterClusterByDis"](dataI, final_cut)
            out[variant + "_filtered"] = boxes(dataI[("chr21", "chr21")]["records"])
            meta[variant] = {"steps": steps, "final_cut": final_cut, "max_cut": int(np.max(cuts)),
                             "n_combined": n_before, "n_filtered": len(dataI[("chr21", "chr21")]["records"])}
            print(variant, {k: v for k, v in meta[variant].items() if k != "steps"}, [s.get("cut_out") for s in steps])
    np.savez_compressed(os.path.join(HERE, "chr21_pipe.npz"), **out)
    with open(os.path.join(HERE, "chr21_pipe_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
