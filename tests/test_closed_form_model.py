"""CPU: the order-free closed forms (R1-R3) the HIP kernels implement reproduce the
sequential oracle exactly -- including the rarely-firing rules (v2 release, v1 start-point
steal, v1 <minPts drop), which must actually fire somewhere in the case set."""
import numpy as np
import pytest

import cases
import closed_form_model as M
import golden_util as G
import oracle


@pytest.mark.parametrize("family,seed,ncase", [("adversarial", 10, 80), ("plain", 11, 80), ("clumpy", 12, 80)])
def test_v2_closed_form(family, seed, ncase):
    rng = np.random.default_rng(seed)
    gen = getattr(cases, family + "_case")
    released = unc = 0
    for k in range(ncase):
        ids, X, Y, eps, minPts = gen(rng, k)
        st = {}
        got = M.labels_v2(X, Y, eps, minPts, st)
        assert np.array_equal(got, oracle.labels("v2", X, Y, eps, minPts)), (family, k)
        released += st.get("released", 0)
        unc += st.get("uncertain", 0)
        assert st.get("max_adj", 0) <= 4          # geometric bound the kernels rely on
    if family != "clumpy":
        assert released > 0 and unc >= released


@pytest.mark.parametrize("family,seed,ncase", [("adversarial", 20, 80), ("plain", 21, 80), ("clumpy", 22, 80)])
def test_v1_closed_form(family, seed, ncase):
    rng = np.random.default_rng(seed)
    gen = getattr(cases, family + "_case")
    steals = dropped = 0
    for k in range(ncase):
        ids, X, Y, eps, minPts = gen(rng, k)
        st = {}
        got = M.labels_v1(X, Y, eps, minPts, st)
        assert np.array_equal(got, oracle.labels("v1", X, Y, eps, minPts)), (family, k)
        steals += st["steals"] if "steals" in st else 0
        dropped += st["dropped"] if "dropped" in st else 0
    if family != "clumpy":
        assert steals > 0


@pytest.mark.parametrize("family,seed,ncase", [("adversarial", 30, 40), ("plain", 31, 40), ("clumpy", 32, 40)])
def test_block_closed_form(family, seed, ncase):
    rng = np.random.default_rng(seed)
    gen = getattr(cases, family + "_case")
    for k in range(ncase):
        ids, X, Y, eps, minPts = gen(rng, k)
        assert np.array_equal(M.labels_block(X, Y, eps, minPts), oracle.labels("block", X, Y, eps, minPts)), (family, k)


def test_closed_forms_on_chr21():
    X, Y = G.chr21_xy()
    st = {}
    assert np.array_equal(M.labels_v2(X, Y, 2000, 5, st), G.chr21_labels("v2", 2000, 5))
    st1 = {}
    assert np.array_equal(M.labels_v1(X, Y, 2000, 5, st1), G.chr21_labels("v1", 2000, 5))


def test_v1_closed_form_on_scaled_coordinates_vs_the_real_class():
    """the axis-stretched use of variant 1 (scripts/callStripes:44-46: X * ext, coordinates beyond int32):
    closed form R1 on int64 coordinates == the real cLoops.cDBSCAN on the scaled matrix (runs where
    /root/reference exists); it is the checker of tests/test_gpu_weighted.py"""
    import refload
    if not refload.available():
        pytest.skip("reference checkout not present")
    rng = np.random.default_rng(3)
    for k in range(8):
        n = int(rng.integers(200, 1500))
        X = rng.integers(10 ** 8, 10 ** 8 + 40000, n).astype(np.int64)
        Y = X + rng.integers(0, 400000, n)
        if k % 2:
            X[: n // 3] = X[0] + rng.integers(-3, 4, n // 3)      # a vertical stripe
        ids = np.arange(n, dtype=np.int64)
        for wx, wy in ((50, 1), (1, 50)):
            mat = np.stack([ids, X * wx, Y * wy], 1)
            want = refload.labels_dict_to_array(refload.ref_labels("v1", mat, 20000, 5), ids)
            assert int(mat[:, 1:].max()) > 2 ** 31
            assert np.array_equal(M.labels_v1(X * wx, Y * wy, 20000, 5), want), (k, wx, wy)
