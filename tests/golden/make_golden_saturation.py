#!/usr/bin/env python
"""Golden vectors for the re-sampling flow of scripts/jd2saturation, produced by the script's OWN functions
(generateSamplingData :32-55, its copy of singleDBSCAN / runDBSCAN :56-127, getLoops :154-178; extracted in memory by
tests/refload.py:ref_saturation_namespace and wired to the REAL cDBSCAN class) on the chr21 example -- run in the build
container only:

    python tests/golden/make_golden_saturation.py

np.random.seed(20171012); repeats 2, step 4 (depths 0.25, 0.5, 0.75), eps [1000, 2000], minPts 5, cut 0 and 400 (a larger cut
leaves no self-ligation cluster, every run is skipped and the script dies in `min([])` -- recorded as "error" where it happens).
Stored per sample: a checksum and the first rows of the drawn index list, the candidate boxes after the per-eps filter and
combineTwice, the cuts, the minPts the sample was called with."""
import json
import os
import sys
import tempfile
import zlib

import joblib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refload  # noqa: E402
import golden_util as G  # noqa: E402

SEED, REPEATS, STEP, EPS, MINPTS = 20171012, 2, 4, [1000, 2000], 5


def boxes(dataI):
    recs = [r for v in dataI.values() for r in v["records"]]
    return np.asarray([[r[1], r[2], r[4], r[5]] for r in recs], dtype=np.int64).reshape(-1, 4)


def main():
    ns = refload.ref_saturation_namespace()
    X, Y = G.chr21_xy()
    mat = np.stack([np.arange(len(X)), X, Y], 1).astype(np.int64)
    out, meta = {}, {}
    for cut in (0, 400, 3000):
        with tempfile.TemporaryDirectory() as td:
            jd = os.path.join(td, "chr21-chr21.jd")
            joblib.dump(mat, jd)
            fout = os.path.join(td, "sat")
            os.mkdir(fout)
            ns["recorded"][:] = []
            tag = "cut%d" % cut

            def call(f, name, cd):
                """getLoops through the recorder -> dict for the meta file"""
                ns["recorded"][:] = []
                try:
                    ns["getLoops"](f, EPS, MINPTS, 0, cut, os.path.join(td, "o_" + name), cd=cd)
                except ValueError as e:                          # min() of an empty list of cuts: no run passed the skip rule
                    return {"error": "ValueError"}
                dataI, mp, c, _ = ns["recorded"][0]
                out["%s_%s_boxes" % (tag, name)] = boxes(dataI)
                return {"minPts": int(mp), "cut": int(c), "n_boxes": int(len(boxes(dataI)))}
            m = {"full": call(jd, "full", 0), "samples": []}
            np.random.seed(SEED)
            fs = ns["generateSamplingData"](jd, fout, REPEATS, STEP, cut)
            for f in fs:
                rows = joblib.load(f)[:, 0]                      # the ids of the drawn rows (= row numbers of the .jd)
                name = f.split("/")[-2]
                out["%s_%s_rows_head" % (tag, name)] = rows[:64]
                d = call(f, name, 1)
                d.update({"name": name, "n_rows": int(len(rows)), "rows_crc32": int(zlib.crc32(np.ascontiguousarray(rows, dtype=np.int64).tobytes()))})
                m["samples"].append(d)
            meta[tag] = m
            print(tag, m["full"], [(s["name"], s["n_rows"], s.get("minPts"), s.get("cut"), s.get("n_boxes"), s.get("error")) for s in m["samples"]])
    np.savez_compressed(os.path.join(HERE, "chr21_saturation.npz"), **out)
    with open(os.path.join(HERE, "chr21_saturation_meta.json"), "w") as fh:
        json.dump({"seed": SEED, "repeats": REPEATS, "step": STEP, "eps": EPS, "minPts": MINPTS, "runs": meta}, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
