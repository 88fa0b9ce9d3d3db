"""Pins the C oracle against the REAL reference classes (only where /root/reference exists).

Plain equality of label arrays (ids included) on the adversarial families of SURVEY.md
Appendix A plus a clumpy family, for all three variants."""
import numpy as np
import pytest

import refload
import cases
import oracle

pytestmark = pytest.mark.skipif(not refload.available(), reason="reference checkout not present")


def _check(variant, ids, X, Y, eps, minPts):
    mat = np.stack([ids, X, Y], 1)
    ref = refload.labels_dict_to_array(refload.ref_labels(variant, mat, eps, minPts), ids)
    got = oracle.labels(variant, X, Y, eps, minPts)
    assert np.array_equal(ref, got), (variant, eps, minPts, len(X), int((ref != got).sum()))
    return ref


@pytest.mark.parametrize("variant", ["v1", "v2", "block"])
@pytest.mark.parametrize("family,seed,ncase", [("adversarial", 0, 120), ("plain", 1, 120), ("clumpy", 2, 120)])
def test_oracle_matches_reference(variant, family, seed, ncase):
    rng = np.random.default_rng(seed)
    gen = getattr(cases, family + "_case")
    nclustered = 0
    for k in range(ncase):
        ids, X, Y, eps, minPts = gen(rng, k)
        ref = _check(variant, ids, X, Y, eps, minPts)
        nclustered += int((ref >= 0).sum())
    assert nclustered > 0
