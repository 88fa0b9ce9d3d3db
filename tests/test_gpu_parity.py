"""GPU parity tests: the HIP path (through the C ABI) against the committed golden vectors
(made from the REAL reference) and against the CPU oracle on seeded inputs.  Bit-exact."""
import numpy as np
import pytest

import golden_util as G
import oracle
from cloops_amd import api

pytestmark = pytest.mark.gpu

ROT = ["v2", "v1", "block"]


def gpu_labels(variant, X, Y, eps, minPts, cut=0):
    ch = api.Chromosome(X, Y)
    try:
        return ch.cluster(variant, eps, minPts, cut)
    finally:
        ch.close()


def test_library_loaded_and_device_present():
    assert api.device_count() >= 1


@pytest.mark.parametrize("eps", [7, 100, 2000])
def test_neighbor_counts_bruteforce(eps):
    rng = np.random.default_rng(eps)
    n = 3000
    X = rng.integers(0, 40 * eps, n)
    Y = X + rng.integers(0, 30 * eps, n)
    ch = api.Chromosome(X, Y)
    got = ch.neighbor_counts(eps)
    ch.close()
    assert np.array_equal(got, oracle.neighbor_counts(X, Y, eps))


@pytest.mark.parametrize("variant", ROT)
@pytest.mark.parametrize("family", ["adversarial", "plain", "clumpy"])
def test_families_golden(variant, family):
    for k, ids, X, Y, eps, minPts, gold in G.family_cases(family):
        res = gpu_labels(variant, X, Y, eps, minPts)
        assert np.array_equal(res.labels, gold[variant]), (family, k, variant, eps, minPts)


@pytest.mark.parametrize("variant", ROT)
@pytest.mark.parametrize("eps,minPts", [(500, 5), (1000, 5), (2000, 5), (5000, 20)])
def test_chr21_golden(variant, eps, minPts):
    X, Y = G.chr21_xy()
    res = gpu_labels(variant, X, Y, eps, minPts)
    gold = G.chr21_labels(variant, eps, minPts)
    assert np.array_equal(res.labels, gold)
    m = G.meta()["chr21_%s_%d_%d" % (variant, eps, minPts)]
    assert res.n_clusters == m["clusters"]
    # cluster table against a host recomputation from the labels
    for c in np.unique(gold[gold >= 0])[:50]:
        sel = gold == c
        b = res.boxes[c]
        assert (b["count"], b["min_x"], b["max_x"], b["min_y"], b["max_y"]) == (
            sel.sum(), X[sel].min(), X[sel].max(), Y[sel].min(), Y[sel].max())


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_chr21_chain_with_cut(variant):
    """config 1 (-m 1): the cut pre-filter of pipe.py:59-63 runs on the GPU."""
    X, Y = G.chr21_xy()
    for step in G.meta()["chr21_chain_" + variant]:
        res = gpu_labels(variant, X, Y, step["eps"], step["minPts"], step["cut_in"])
        assert np.array_equal(res.labels, G.chr21_chain_labels(variant, step["eps"])), step
        assert res.n_clusters == step["clusters"]


@pytest.mark.parametrize("variant", ROT)
def test_synth150k_golden(variant):
    X, Y, z = G.synth150k()
    for eps, minPts in ((2000, 5), (5000, 20)):
        res = gpu_labels(variant, X, Y, eps, minPts)
        assert np.array_equal(res.labels, z["%s_%d_%d" % (variant, eps, minPts)])


@pytest.mark.parametrize("variant", ROT)
def test_oracle_midsize(variant):
    """2 M synthetic PETs against the C oracle (sizes the oracle finishes in seconds)."""
    from cloops_amd.synth import synth_chrom
    X, Y = synth_chrom(2000000, 248956422, 77)
    for eps, minPts in ((2000, 5), (5000, 20)):
        res = gpu_labels(variant, X, Y, eps, minPts)
        want = oracle.labels(variant, X, Y, eps, minPts)
        assert np.array_equal(res.labels, want), (variant, eps, minPts, int((res.labels != want).sum()))


@pytest.mark.parametrize("variant", ROT)
def test_cut_filter_matches_oracle(variant):
    """cut > 0 for every variant (block has no golden chain): GPU filter == pipe.py:59-63."""
    X, Y = G.chr21_xy()
    for eps, minPts, cut in ((1000, 5, 4601), (2000, 5, 13532)):
        res = gpu_labels(variant, X, Y, eps, minPts, cut)
        want = oracle.single_dbscan(variant, X, Y, eps, minPts, cut)["labels"]
        assert np.array_equal(res.labels, want), (variant, eps, cut)


def test_async_two_in_flight_matches_sync():
    """cl_cluster_async / cl_wait with two runs in flight give the synchronous results."""
    X, Y = G.chr21_xy()
    ch = api.Chromosome(X, Y)
    try:
        params = [("v2", 500, 5, 0), ("v1", 2000, 5, 0), ("block", 1000, 5, 0), ("v2", 1000, 5, 4601), ("v2", 5000, 20, 0)]
        sync = [ch.cluster(v, e, m, c) for v, e, m, c in params]
        got = []
        ch.cluster_async(*params[0])
        for k in range(len(params)):
            if k + 1 < len(params):
                ch.cluster_async(*params[k + 1])
            got.append(ch.wait(copy=True))
        for a, b in zip(sync, got):
            assert np.array_equal(a.labels, b.labels)
            assert a.n_clusters == b.n_clusters and a.max_label == b.max_label
            assert np.array_equal(a.boxes, b.boxes)
        with pytest.raises(Exception):
            ch.cluster_async("v2", 500, 5); ch.cluster_async("v2", 500, 5); ch.cluster_async("v2", 500, 5)
        ch.wait(); ch.wait()
    finally:
        ch.close()


@pytest.mark.parametrize("variant", ROT)
def test_dense400k_headline_regime(variant):
    """labels made by the REAL classes at the headline's density and settings (tests/golden/make_golden_dense.py), through the
    sweep's own path: both lists announced in one call (cl_sweep_plan), traversal level 4, the runs in the sweep's order -- so
    that the words of an eps are made by the sorted-key region query under the minPts list and the cut band is re-queried"""
    X, Y, z, m = G.dense400k()
    ch = api.Chromosome(X, Y)
    try:
        if variant != "block":
            ch.sweep_plan([5000, 7500, 10000], [50, 40, 30, 20])
        for eps, minPts, cut in G.DENSE_SETTINGS:
            key = "%s_%d_%d_%d" % (variant, eps, minPts, cut)
            res = ch.cluster(variant, eps, minPts, cut)
            assert np.array_equal(res.labels, z[key]), key
            assert res.n_clusters == m["runs"][key]["clusters"]
            if variant != "block":
                # a later run of the same eps under another cut reads the cached words + its cut band: against the oracle
                res2 = ch.cluster(variant, eps, 20, cut + 700)
                assert ch.last_region_mode() == 2
                assert np.array_equal(res2.labels, oracle.single_dbscan(variant, X, Y, eps, 20, cut + 700)["labels"]), (key, "re-use")
    finally:
        ch.close()
