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


@pytest.mark.skipif(not __import__("refload").available(), reason="reference checkout not present")
def test_table_equals_reference_functions(tmp_path):
    """cross-check against the reference's OWN getIntSig / markIntSig (imported from /root/reference through
    tests/refload.py, py2 -> py3 patched in memory): same table, cell by cell"""
    import refload
    ns = refload.ref_cmodel_namespace()
    z, meta = pipe_checks.pipe_golden()
    X, Y = G.chr21_xy()
    mat = np.stack([np.arange(len(X)), X, Y], 1).astype(np.int64)
    f = os.path.join(str(tmp_path), "chr21-chr21.jd")
    joblib.dump(mat, f)
    recs = [["chr21", int(a), int(b), "chr21", int(c), int(d)] for a, b, c, d in z["v2_filtered"][:150]]
    want = ns["markIntSig"](ns["getIntSig"](f, [list(r) for r in recs], [5], 0))
    got = cModel.markIntSig(cModel.getIntSig(f, recs, [5], 0))
    assert list(got.index) == list(want.index) and list(got.columns) == list(want.columns)
    for col in want.columns:
        assert got[col].tolist() == want[col].tolist(), col


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


def test_windows_floor_semantics():
    """the 22 windows of a record: Python-2 floor arithmetic of getNearbyPairRegions (cModel.py:89-93)"""
    iva, ivb, dist, w = cModel._windows([["c", 101, 204, "c", 1001, 1104]])
    lo, hi = w[0, :22], w[0, 22:]
    # ca = 305 // 2 = 152, sa = 103 // 2 = 51, step = (51 + 51) // 2 = 51
    assert (lo[0], hi[0], lo[11], hi[11]) == (101, 204, 1001, 1104)
    assert (lo[1], hi[1]) == (max(0, 152 - 5 * 51 - 51), max(0, 152 - 5 * 51 + 51))
    assert (lo[6], hi[6]) == (152 + 51 - 51, 152 + 51 + 51)
    assert dist[0] == abs((1001 + 1104) / 2.0 - (101 + 204) / 2.0)


def test_counts_are_inclusive_and_use_row_positions():
    import fake_backend
    X = np.array([100, 150, 200, 100])
    Y = np.array([500, 100, 150, 900])
    m = fake_backend.IndexSets(X, Y)
    assert m.side(100, 150, 0).tolist() == [0, 1, 3]       # X in [100,150]
    assert m.side(100, 150, 1).tolist() == [1, 2]          # Y in [100,150]
    assert m.region(100, 150).tolist() == [0, 1, 2, 3]
    w = np.zeros((1, 44), np.int32)
    w[0, 0], w[0, 22], w[0, 11], w[0, 33] = 100, 150, 500, 900
    c, n = fake_backend.FakeChromosome(X, Y).sig_counts(w)
    assert (c[0, 0], c[0, 11], c[0, 22], n) == (4, 2, 2, 4)      # ra, rb, rab (cModel.py:73-80), N


def _one_end(xa, xb, ya, yb):
    """interval overlap as the reference spells it (checkOneEndOverlap): one end of either inside the other"""
    return ya <= xa <= yb or ya <= xb <= yb or xa <= ya <= xb or xa <= yb <= xb


def test_overlap_lists_match_reference_predicate():
    """the sweep-based candidate enumeration of the duplicate removal == the brute-force pair predicate (cModel.py:174-195)"""
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
                    if chrom[i] == chrom[j] and _one_end(iv[i, 0], iv[i, 1], iv[j, 0], iv[j, 1])
                    and _one_end(iv[i, 2], iv[i, 3], iv[j, 2], iv[j, 3])]
            assert adj[i].tolist() == want
