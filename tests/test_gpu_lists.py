"""GPU: the list form of K3 / K4 / K5 (cloops_amd/csrc/k_lists.hip, cl_set_traversal).

The reference expands clusters from core points only (cDBSCAN2.py:114-192 queryGrid, cDBSCAN.py:155-184 expandCluster); the
library compacts a run's cores and its non-core PETs with a neighbour into two lists behind the region query (level 3), at
level 4 straight from the base layout with the cut applied by index (no copy of the layout, count cache in base-position
space).  Levels 0 .. 2 keep the LDS-tile kernels over every PET for the components / the border rule / the labels.  Every level
must give the labels of the sequential oracle -- with and without a cut, re-used counts included, both rotated variants."""
import numpy as np
import pytest

import oracle
from cloops_amd import api
from cloops_amd.synth import synth_chrom

pytestmark = pytest.mark.gpu

N = 150000


def _want(variant, X, Y, eps, m, cut):
    keep = Y.astype(np.int64) - X >= cut
    want = np.full(len(X), -1, np.int32)
    want[keep] = oracle.labels(variant, X[keep], Y[keep], eps, m)
    return want


@pytest.mark.parametrize("variant", ["v2", "v1"])
@pytest.mark.parametrize("level", [0, 1, 2, 3, 4])
def test_levels_equal_oracle(variant, level):
    X, Y = synth_chrom(N, 46709983, 23)
    ch = api.Chromosome(X, Y)
    ch.set_traversal(level)
    try:
        for eps, m, cut in ((2000, 5, 0), (5000, 20, 0), (5000, 10, 3000), (5000, 10, 4500), (5000, 12, 2500), (500, 3, 0), (500, 3, 700)):
            got = ch.cluster(variant, eps, m, cut)
            want = _want(variant, X, Y, eps, m, cut)
            assert np.array_equal(got.labels, want), (variant, level, eps, m, cut, int((got.labels != want).sum()))
            assert got.n_clusters == len(np.unique(want[want >= 0]))
    finally:
        ch.close()


def test_level4_makes_no_copy_for_reusing_runs():
    """at level 4 the words of an eps live in base-position space: a run under a cut re-uses them (mode 2: band query) even when it is
    the first run of the eps, and a run without a cut takes them as they are (mode 1)"""
    X, Y = synth_chrom(N, 46709983, 29)
    ch = api.Chromosome(X, Y)
    ch.set_count_thresholds([5, 10, 20])
    try:
        modes = []
        runs = ((5000, 20, 3000),      # makes the words of the eps: on the base layout (PETs below its cut skipped) + its own cut band
                (5000, 10, 3500),      # re-uses them: band query only
                (5000, 10, 3000),      # the making run's cut again: still a band query (its band counted removed neighbours)
                (7500, 20, 0),         # another eps, no cut: the query on the base layout
                (7500, 10, 0),         # same layout, no cut: the words as they are
                (7500, 5, 2000))       # a cut on top of them: band query
        for eps, m, cut in runs:
            got = ch.cluster("v2", eps, m, cut)
            modes.append(ch.last_region_mode())
            assert np.array_equal(got.labels, _want("v2", X, Y, eps, m, cut)), (eps, m, cut)
        if api.TRAVERSAL_OVERRIDE in (None, 4, "4"):         # (the suite run under CLOOPS_TRAVERSAL < 4 checks the labels only: the modes are level 4's)
            assert modes == [0, 2, 2, 0, 1, 2], modes
    finally:
        ch.close()


def test_sweep_step_statistics_from_the_lists():
    """the distance statistics of a sweep step read the run's lists (K7, sorted == 2): the chained cut equals the oracle's"""
    from cloops_amd import pipe, ests
    X, Y = synth_chrom(120000, 46709983, 31)
    pipe.CACHE.clear()
    f = pipe.CACHE.put_arrays("chrL-chrL", X, Y)
    try:
        dataI, cut, cuts, steps = pipe.runSweepFast([f], [1500, 3000], [8, 5], cut=0)
        c = 0
        for st in steps:
            ref = oracle.single_dbscan("v2", X, Y, st["eps"], st["minPts"], c)
            assert st["n_inter"] == len(ref["dataI"]) and st["n_self"] == len(ref["dataS"])
            c = ests.estIntSelCutFrag(ref["dis"], ref["dss"])[0]
            assert st["cut_out"] == c
    finally:
        pipe.CACHE.clear()


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_pairs_form_equals_row_aligned_labels(variant):
    """cl_cluster_pairs_async: the labels as the reference holds them -- (row, label) of the clustered points only
    (cDBSCAN2.py:186-191) -- reassemble to exactly the row-aligned labels of cl_cluster, run after run on two result slots"""
    if api.TRAVERSAL_OVERRIDE is not None and int(api.TRAVERSAL_OVERRIDE) < 3:
        pytest.skip("the pairs form needs the list form of a run (traversal level >= 3)")
    X, Y = synth_chrom(N, 46709983, 37)
    ref = api.Chromosome(X, Y)
    ch = api.Chromosome(X, Y)
    try:
        pending = []
        for eps, m, cut in ((2000, 5, 0), (5000, 20, 0), (5000, 10, 3000), (5000, 10, 4500)):
            ch.cluster_pairs_async(variant, eps, m, cut)
            pending.append((eps, m, cut))
            if len(pending) == 2:
                e0, m0, c0 = pending.pop(0)
                res, pairs = ch.wait_pairs(copy=True)
                want = ref.cluster(variant, e0, m0, c0)
                got = np.full(len(X), -1, np.int32)
                got[pairs[:, 0]] = pairs[:, 1]
                assert len(np.unique(pairs[:, 0])) == len(pairs) == int((want.labels >= 0).sum())
                assert np.array_equal(got, want.labels) and res.n_clusters == want.n_clusters
        while pending:
            e0, m0, c0 = pending.pop(0)
            res, pairs = ch.wait_pairs(copy=True)
            want = ref.cluster(variant, e0, m0, c0)
            got = np.full(len(X), -1, np.int32)
            got[pairs[:, 0]] = pairs[:, 1]
            assert np.array_equal(got, want.labels)
    finally:
        ch.close()
        ref.close()


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_rowmask_form_equals_row_aligned_labels(variant):
    """cl_cluster_rowmask_async: one bit per row + the labels of the set rows in ascending row order -- the clustered points of
    the reference's `.labels` (cDBSCAN2.py:186-191) -- reassemble to exactly the row-aligned labels of cl_cluster, run after run
    on two result slots, with the copy deferred (cl_set_pairs_defer / cl_pairs_sync) and not"""
    if api.TRAVERSAL_OVERRIDE is not None and int(api.TRAVERSAL_OVERRIDE) < 3:
        pytest.skip("the row-mask form needs the list form of a run (traversal level >= 3)")
    X, Y = synth_chrom(N + 37, 46709983, 41)               # (a last mask word that is not full)
    ref = api.Chromosome(X, Y)
    ch = api.Chromosome(X, Y)

    def check(setting, defer):
        e0, m0, c0 = setting
        res, mask, labels = ch.wait_rowmask(defer=defer)
        if defer:
            ch.pairs_sync()
        want = ref.cluster(variant, e0, m0, c0)
        rows = ch.rows_of_mask(mask)
        assert len(mask) == (len(X) + 63) // 64
        assert np.array_equal(rows, np.flatnonzero(want.labels >= 0)), setting
        assert np.array_equal(labels, want.labels[rows]), setting
        assert res.n_clusters == want.n_clusters
    try:
        pending = []
        for k, setting in enumerate(((2000, 5, 0), (5000, 20, 0), (5000, 10, 3000), (5000, 10, 4500), (800, 3, 0))):
            ch.cluster_rowmask_async(variant, *setting)
            pending.append(setting)
            if len(pending) == 2:
                check(pending.pop(0), defer=bool(k & 1))
        while pending:
            check(pending.pop(0), defer=False)
    finally:
        ch.close()
        ref.close()


def test_rowmask_form_on_tiny_and_empty_results():
    """fewer rows than one mask word; a run that clusters nothing (all bits clear, no labels); a capacity below the labelled count
    is an argument error at cl_wait, not an overrun"""
    if api.TRAVERSAL_OVERRIDE is not None and int(api.TRAVERSAL_OVERRIDE) < 3:
        pytest.skip("the row-mask form needs the list form of a run (traversal level >= 3)")
    import ctypes
    from cloops_amd import _lib
    rng = np.random.default_rng(5)
    X = np.sort(rng.integers(1000, 3000, 40)).astype(np.int32)
    Y = (X + rng.integers(100, 400, 40)).astype(np.int32)
    ch = api.Chromosome(X, Y)
    try:
        want = ch.cluster("v2", 500, 3, 0)
        ch.cluster_rowmask_async("v2", 500, 3, 0)
        res, mask, labels = ch.wait_rowmask()
        rows = ch.rows_of_mask(mask)
        assert len(mask) == 1 and np.array_equal(rows, np.flatnonzero(want.labels >= 0)) and np.array_equal(labels, want.labels[rows])
        assert len(rows) > 0
        ch.cluster_rowmask_async("v2", 1, 30, 0)               # nothing can cluster
        res, mask, labels = ch.wait_rowmask()
        assert res.n_clusters == 0 and int(mask[0]) == 0 and len(labels) == 0
        # capacity below the labelled count: CL_ERR_ARG at cl_wait
        lib = _lib.load()
        p = lib.cl_host_alloc(8 + 4 * 40)
        try:
            _lib.check(lib.cl_cluster_rowmask_async(ch._h, api.VARIANTS["v2"], 500, 3, 0, ctypes.c_void_p(p), 1))
            nc, ml = ctypes.c_int32(0), ctypes.c_int32(-1)
            rc = lib.cl_wait(ch._h, ctypes.byref(nc), ctypes.byref(ml))
            assert rc != 0 and b"capacity_labels" in lib.cl_last_error()
        finally:
            lib.cl_host_free(ctypes.c_void_p(p))
        # the handle still works
        again = ch.cluster("v2", 500, 3, 0)
        assert np.array_equal(again.labels, want.labels)
    finally:
        ch.close()
