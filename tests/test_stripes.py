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


def test_nearby_stripes_floor_semantics():
    ivas, ivbs = stripes.getNearbyStripes([100, 205], [1000, 1011])          # a longer: b slides by sb = 5
    assert ivas == [[100, 205]] * 10
    assert ivbs[0] == [1005 - 25 - 5, 1005 - 25 + 5] and ivbs[-1] == [1005 + 25 - 5, 1005 + 25 + 5]
    ivas, ivbs = stripes.getNearbyStripes([10, 21], [1000, 2000])            # b longer: a slides, clipped at 0
    assert ivbs == [[1000, 2000]] * 10 and ivas[0] == [0, 0] and ivas[5] == [15 + 5 - 5, 15 + 5 + 5]
    assert stripes.getNearbyStripes([0, 10], [100, 110]) is None
