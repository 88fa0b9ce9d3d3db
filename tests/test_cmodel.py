"""CPU: significance + `.loop` rows (cloops_amd/cModel.py) against the golden tables made by the
(py2->py3 converted) reference cModel on config 1 -- text-identical `.loop` files."""
import os

import joblib
import numpy as np
import pandas as pd
import pytest

import golden_util as G
import pipe_checks
from cloops_amd import cModel


@pytest.fixture(autouse=True)
def cpu_backend(monkeypatch):
    """host logic on CPU: the GPU counting kernel is replaced by its numpy stand-in"""
    import fake_backend
    from cloops_amd import api, pipe
    monkeypatch.setattr(api, "Chromosome", fake_backend.FakeChromosome)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    yield
    pipe.CACHE.clear()


def test_set_based_restatement_identical(tmp_path):
    """the literal set-based restatement (getIntSigFromMat) also reproduces the golden table"""
    z, meta = pipe_checks.pipe_golden()
    X, Y = G.chr21_xy()
    mat = np.stack([np.arange(len(X)), X, Y], 1).astype(np.int64)
    recs = [["chr21", int(a), int(b), "chr21", int(c), int(d)] for a, b, c, d in z["v2_filtered"]]
    ds = cModel.markIntSig(cModel.getIntSigFromMat(mat, recs, [5], 0))
    out = os.path.join(str(tmp_path), "x.loop")
    ds.to_csv(out, sep="\t", index_label="loopId")
    assert open(out).read() == open(os.path.join(G.GOLD, "chr21_v2.loop")).read()


@pytest.mark.parametrize("variant", ["v2", "v1"])
@pytest.mark.parametrize("hic", [0, 1])
def test_loop_file_identical(variant, hic, tmp_path):
    z, meta = pipe_checks.pipe_golden()
    X, Y = G.chr21_xy()
    mat = np.stack([np.arange(len(X)), X, Y], 1).astype(np.int64)
    f = os.path.join(str(tmp_path), "chr21-chr21.jd")
    joblib.dump(mat, f)
    recs = [["chr21", int(a), int(b), "chr21", int(c), int(d)] for a, b, c, d in z[variant + "_filtered"]]
    dataI = {("chr21", "chr21"): {"f": f, "records": recs}}
    fout = os.path.join(str(tmp_path), "out")
    assert cModel.runStat(dataI, [5], 0, 1, fout, hichip=hic) == 0        # pipe.py:284 passes cut = 0
    got = open(fout + ".loop").read()
    want = open(os.path.join(G.GOLD, "chr21_%s%s.loop" % (variant, "_hic" if hic else ""))).read()
    assert got == want
    df = pd.read_csv(fout + ".loop", sep="\t", index_col=0)
    assert int(df["significant"].sum()) == (252 if hic else 202)


def test_nearby_regions_floor_semantics():
    ivas, ivbs = cModel.getNearbyPairRegions([101, 204], [1001, 1104])
    # ca = 305 // 2 = 152, sa = 103 // 2 = 51, step = (51 + 51) // 2 = 51
    assert ivas[0] == [max(0, 152 - 5 * 51 - 51), max(0, 152 - 5 * 51 + 51)] and len(ivas) == 10 and len(ivbs) == 10
    assert ivas[5] == [152 + 51 - 51, 152 + 51 + 51]


def test_counts_are_inclusive_and_use_row_positions():
    mat = np.array([[10, 100, 500], [11, 150, 100], [12, 200, 150], [13, 100, 900]], dtype=np.int64)
    m = cModel.CoverageModel(mat)
    assert m.side([100, 150], 0).tolist() == [0, 1, 3]       # X in [100,150]
    assert m.side([100, 150], 1).tolist() == [1, 2]          # Y in [100,150]
    assert m.region([100, 150]).tolist() == [0, 1, 2, 3]
    assert cModel.getPETsforRegions([100, 150], [500, 900], m) == (4, 2, 2)


def test_overlap_lists_match_reference_predicate():
    """the sweep-based candidate enumeration of removeDup == brute-force checkOverlap (cModel.py:174-195)"""
    rng = np.random.default_rng(0)
    for t in range(12):
        L = int(rng.integers(1, 300))
        a0 = rng.integers(0, 20000, L)
        a1 = a0 + rng.integers(0, 300, L)
        b0 = a0 + rng.integers(0, 5000, L)
        b1 = b0 + rng.integers(0, 300, L)
        if t % 3 == 0 and L >= 3:
            a1[:3] = a0[:3] + rng.integers(5000, 30000, 3)          # a few very long anchors
        iv = np.stack([a0, a1, b0, b1], 1).astype(np.int64)
        chrom = np.asarray(["c1|c1" if x else "c2|c2" for x in rng.random(L) < 0.8])
        adj = cModel._overlap_lists(iv, chrom)
        for i in range(L):
            want = [j for j in range(i + 1, L)
                    if cModel.checkOverlap([chrom[i], iv[i, 0], iv[i, 1]], [chrom[i], iv[i, 2], iv[i, 3]],
                                           [chrom[j], iv[j, 0], iv[j, 1]], [chrom[j], iv[j, 2], iv[j, 3]])]
            assert adj[i].tolist() == want
