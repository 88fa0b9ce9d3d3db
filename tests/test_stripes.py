"""CPU: the stripe significance of scripts/callStripes (cloops_amd/stripes.py) against golden `.stripe` tables
made with the reference's own functions (tests/golden/make_golden_stripe_table.py) -- text-identical."""
import json
import os

import numpy as np
import pandas as pd
import pytest

import golden_util as G
from cloops_amd import stripes, pipe


def records_from_labels(lab, X, Y):
    out = []
    for c in np.unique(lab[lab >= 0]):
        sel = lab == c
        out.append(["chr21", int(X[sel].min()), int(X[sel].max()), "chr21", int(Y[sel].min()), int(Y[sel].max()), int(sel.sum())])
    return out


@pytest.fixture()
def chr21(monkeypatch):
    import fake_backend
    from cloops_amd import api
    monkeypatch.setattr(api, "Chromosome", fake_backend.FakeChromosome)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    X, Y = G.chr21_xy()
    f = pipe.CACHE.put_arrays("chr21-chr21", X, Y)
    yield f, X, Y
    pipe.CACHE.clear()


@pytest.mark.parametrize("name", ["x_horizontal", "y_vertical"])
def test_stripe_table_identical(chr21, name, tmp_path):
    f, X, Y = chr21
    meta = json.load(open(os.path.join(G.GOLD, "chr21_stripes_meta.json")))
    lab = np.load(os.path.join(G.GOLD, "chr21_stripes_labels.npz"))["x50" if name.startswith("x") else "y50"]
    dataI = records_from_labels(lab, X, Y)                 # what singleStripDBSCAN returns (GPU test checks that part)
    assert len(dataI) == meta[name + "_clusters"]
    ds = stripes.filterCandidateStripes({("chr21", "chr21"): dataI}, pets=meta["pets"], lengthFoldDiff=meta["lengthFoldDiff"])
    cand = ds[("chr21", "chr21")]
    assert len(cand) == meta[name + "_candidates"]
    tab = stripes.markStripeSig(pd.concat([stripes.estStripeSig(f, cand)]))
    out = os.path.join(str(tmp_path), "o.stripe")
    tab.to_csv(out, sep="\t", index_label="stripeId")
    assert open(out).read() == open(os.path.join(G.GOLD, "chr21_%s.stripe" % name)).read()
    assert int(tab["significant"].sum()) == meta[name + "_significant"]


def test_stripe_windows_floor_semantics():
    """background windows (scripts/callStripes:89-120, py2 floor arithmetic): the longer anchor stays, the shorter
    slides by its half-length; equal lengths fail like the script does"""
    iva, ivb, w = stripes._stripe_windows([["c", 100, 205, "c", 1000, 1011], ["c", 10, 21, "c", 1000, 2000]])
    lo, hi = w[:, :22], w[:, 22:]
    assert lo[0, :11].tolist() == [100] * 11 and hi[0, :11].tolist() == [205] * 11          # a longer: stays
    assert (lo[0, 12], hi[0, 12]) == (1005 - 25 - 5, 1005 - 25 + 5) and (lo[0, 21], hi[0, 21]) == (1005 + 25 - 5, 1005 + 25 + 5)
    assert lo[1, 11:].tolist() == [1000] * 11 and hi[1, 11:].tolist() == [2000] * 11        # b longer: stays
    assert (lo[1, 1], hi[1, 1]) == (0, 0) and (lo[1, 6], hi[1, 6]) == (15 + 5 - 5, 15 + 5 + 5)   # clipped at 0
    with pytest.raises(TypeError):
        stripes._stripe_windows([["c", 0, 10, "c", 100, 110]])
