"""Shared assertions: cloops_amd.pipe against the golden vectors produced by the
reference's own dispatch functions (tests/golden/make_golden_pipe.py)."""
import contextlib
import io
import json
import os

import joblib
import numpy as np

import golden_util as G


def pipe_golden():
    z = np.load(os.path.join(G.GOLD, "chr21_pipe.npz"))
    with open(os.path.join(G.GOLD, "chr21_pipe_meta.json")) as fh:
        return z, json.load(fh)


def write_chr21_jd(tmpdir):
    X, Y = G.chr21_xy()
    mat = np.stack([np.arange(len(X)), X, Y], 1).astype(np.int64)
    f = os.path.join(str(tmpdir), "chr21-chr21.jd")
    joblib.dump(mat, f)
    return f


def boxes(records):
    return np.asarray([[r[1], r[2], r[4], r[5]] for r in records], dtype=np.int64).reshape(-1, 4)


def check_run_dbscan_chain(pipe, f, variant="v2"):
    z, meta = pipe_golden()
    for st in meta[variant]["steps"]:
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            dataI, dataS, dis, dss = pipe.runDBSCAN([f], st["eps"], st["minPts"], st["cut_in"], 1)
        tag = "%s_%d_%d" % (variant, st["eps"], st["minPts"])
        assert isinstance(dis, list) and isinstance(dss, list)
        assert list(dataI.keys()) == [("chr21", "chr21")] and dataI[("chr21", "chr21")]["f"] == f
        recs = dataI[("chr21", "chr21")]["records"]
        assert recs[0][0] == "chr21" and recs[0][3] == "chr21"
        assert np.array_equal(boxes(recs), z[tag + "_dataI"])            # same boxes, same ORDER
        assert np.array_equal(boxes(dataS), z[tag + "_dataS"])
        assert np.array_equal(np.sort(np.asarray(dis, dtype=np.int64)), z[tag + "_dis_sorted"])
        assert np.array_equal(np.sort(np.asarray(dss, dtype=np.int64)), z[tag + "_dss_sorted"])
        assert err.getvalue() == st["stderr"]                            # the two report lines
        cut2, frags = pipe.estIntSelCutFrag(np.array(dis), np.array(dss))
        assert (cut2, frags) == (st["cut_out"], st["frags"])


def check_sweep(pipe, f, variant="v2"):
    z, meta = pipe_golden()
    dataI, cut, cuts, steps = pipe.runSweep([f], [500, 1000, 2000], [5], cut=0)
    m = meta[variant]
    assert cut == m["final_cut"]
    assert [s.get("cut_out") for s in steps] == [s.get("cut_out") for s in m["steps"]]
    assert np.array_equal(boxes(dataI[("chr21", "chr21")]["records"]), z[variant + "_filtered"])
    assert len(dataI[("chr21", "chr21")]["records"]) == m["n_filtered"]
    dataI2, cut2, _, _ = pipe.runSweep([f], [500, 1000, 2000], [5], cut=0, max_cut=True)
    assert cut2 == m["max_cut"]


def check_sweep_fast(pipe, f, variant="v2"):
    """runSweepFast (GPU-side statistics, array candidates) == the reference's chain."""
    z, meta = pipe_golden()
    m = meta[variant]
    dataI, cut, cuts, steps = pipe.runSweepFast([f], [500, 1000, 2000], [5], cut=0)
    assert cut == m["final_cut"]
    assert [s.get("cut_out") for s in steps] == [s.get("cut_out") for s in m["steps"]]
    assert [s.get("frags") for s in steps] == [s.get("frags") for s in m["steps"]]
    assert np.array_equal(dataI[("chr21", "chr21")]["boxes"], z[variant + "_filtered"])
    _, cut2, _, _ = pipe.runSweepFast([f], [500, 1000, 2000], [5], cut=0, max_cut=True)
    assert cut2 == m["max_cut"]
