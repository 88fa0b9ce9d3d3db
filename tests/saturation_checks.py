"""Shared assertions: cloops_amd.saturation (the re-sampling flow of scripts/jd2saturation) against the golden vectors the script's
own functions produced (tests/golden/make_golden_saturation.py)."""
import json
import os
import zlib

import numpy as np
import pytest

import golden_util as G


def golden():
    z = np.load(os.path.join(G.GOLD, "chr21_saturation.npz"))
    with open(os.path.join(G.GOLD, "chr21_saturation_meta.json")) as fh:
        return z, json.load(fh)


def boxes(dataI):
    recs = [r for v in dataI.values() for r in v["records"]]
    return np.asarray([[r[1], r[2], r[4], r[5]] for r in recs], dtype=np.int64).reshape(-1, 4)


def check_flow(pipe, saturation, cuts=(0, 400, 3000)):
    z, meta = golden()
    X, Y = G.chr21_xy()
    pipe.CACHE.clear()
    jd = pipe.CACHE.put_arrays("chr21-chr21", X, Y)
    try:
        for cut in cuts:
            tag = "cut%d" % cut
            m = meta["runs"][tag]

            def check(f, name, want, cd):
                if "error" in want:
                    with pytest.raises(ValueError):              # min() of an empty list of cuts, like the script
                        saturation.callLoops(f, meta["eps"], meta["minPts"], cut, cd)
                    return
                dataI, c, mp, _ = saturation.callLoops(f, meta["eps"], meta["minPts"], cut, cd)
                assert (mp, c) == (want["minPts"], want["cut"]), (tag, name)
                assert np.array_equal(boxes(dataI), z["%s_%s_boxes" % (tag, name)]), (tag, name)      # same boxes, same ORDER
            check(jd, "full", m["full"], 0)
            np.random.seed(meta["seed"])
            fs = saturation.generateSamplingData(jd, meta["repeats"], meta["step"], cut)
            assert [f.split("/")[-2] for f in fs] == [s["name"] for s in m["samples"]]
            for f, want in zip(fs, m["samples"]):
                r = pipe.CACHE.get(f)
                rows = np.ascontiguousarray(r.ids, dtype=np.int64)            # the drawn rows, in the script's order
                assert len(rows) == want["n_rows"] and zlib.crc32(rows.tobytes()) == want["rows_crc32"]
                assert np.array_equal(rows[:64], z["%s_%s_rows_head" % (tag, want["name"])])
                assert np.array_equal(r.X, X[rows]) and np.array_equal(r.Y, Y[rows])
                assert saturation.sample_depth(f) == float(want["name"].split("_")[1])
                check(f, want["name"], want, 1)
                pipe.CACHE.drop(f)
    finally:
        pipe.CACHE.clear()
