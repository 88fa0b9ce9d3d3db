"""GPU: sorted-layout reuse (cl_set_layout_reuse).  A handle that keeps the sorted arrays of the last eps and starts
further runs from a stream compaction by the cut must give exactly the labels / tables of a handle that sorts every
run for itself, in any order of (eps, minPts, cut) -- and both equal the CPU oracle."""
import numpy as np
import pytest

import golden_util as G
import oracle
from cloops_amd import api
from cloops_amd.synth import synth_chrom

pytestmark = pytest.mark.gpu

# sweeps as pipe() issues them (eps outer loop, minPts inner, the cut changing every run) plus returns to an earlier eps
SEQ = [(500, 5, 0), (500, 4, 4601), (500, 6, 300), (1000, 5, 4601), (1000, 5, 0), (1000, 8, 13532), (2000, 5, 11103),
       (500, 5, 4601), (2000, 3, 10 ** 9), (2000, 5, 0)]


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_reuse_equals_fresh_sort_chr21(variant):
    X, Y = G.chr21_xy()
    a = api.Chromosome(X, Y)
    b = api.Chromosome(X, Y)
    b.set_layout_reuse(False)
    for k, (eps, m, cut) in enumerate(SEQ):
        ra = a.cluster(variant, eps, m, cut)
        rb = b.cluster(variant, eps, m, cut)
        assert np.array_equal(ra.labels, rb.labels), (variant, eps, m, cut)
        assert ra.n_clusters == rb.n_clusters and np.array_equal(ra.boxes, rb.boxes)
        assert a.last_n_in() == b.last_n_in() == int(((Y - X) >= cut).sum())
        if k in (1, 3, 6):
            want = oracle.single_dbscan(variant, X, Y, eps, m, cut)["labels"]
            assert np.array_equal(ra.labels, want)
    a.close()
    b.close()


def test_reuse_async_sweep_dense():
    """the asynchronous form the sweep driver uses, on dense synthetic data with long strips (general sort path)"""
    X, Y = synth_chrom(1500000, 20000000, 77)
    a = api.Chromosome(X, Y)
    b = api.Chromosome(X, Y)
    b.set_layout_reuse(False)
    runs = [(5000, 30, 0), (5000, 20, 4000), (7500, 30, 5200), (7500, 20, 4800), (5000, 30, 0)]
    for eps, m, cut in runs:
        a.cluster_async("v2", eps, m, cut)
        b.cluster_async("v2", eps, m, cut)
        ra, rb = a.wait(copy=True), b.wait(copy=True)
        assert np.array_equal(ra.labels, rb.labels), (eps, m, cut)
        assert np.array_equal(ra.boxes, rb.boxes)
    want = oracle.single_dbscan("v2", X, Y, 7500, 20, 4800)["labels"]
    a.cluster_async("v2", 7500, 20, 4800)
    assert np.array_equal(a.wait().labels, want)
    a.close()
    b.close()


def test_neighbor_counts_with_reuse():
    X, Y = synth_chrom(200000, 5000000, 5)
    a = api.Chromosome(X, Y)
    b = api.Chromosome(X, Y)
    b.set_layout_reuse(False)
    for eps, cut in ((1500, 0), (1500, 2500), (3000, 2500)):
        assert np.array_equal(a.neighbor_counts(eps, cut), b.neighbor_counts(eps, cut))
    a.close()
    b.close()


@pytest.mark.parametrize("variant", ["v2", "v1"])
@pytest.mark.parametrize("reuse", [True, False])
def test_sort_index_modes_agree(variant, reuse):
    """cl_set_sort_index: layouts taken from the q index (a stable strip sort of the rows kept sorted by q) equal the
    layouts of the full (strip, q) sort -- never / built at the second sort / built at the first -- and the oracle"""
    X, Y = synth_chrom(300000, 8000000, 11)
    hs = []
    for mode in (-1, 0, 1):
        h = api.Chromosome(X, Y)
        h.set_sort_index(mode)
        h.set_layout_reuse(reuse)
        hs.append(h)
    seq = [(1500, 5, 0), (1500, 4, 2500), (3000, 6, 2500), (700, 3, 0), (3000, 5, 9000), (1500, 5, 0)]
    for k, (eps, m, cut) in enumerate(seq):
        rs = [h.cluster(variant, eps, m, cut) for h in hs]
        for r in rs[1:]:
            assert np.array_equal(r.labels, rs[0].labels), (variant, reuse, eps, m, cut)
            assert np.array_equal(r.boxes, rs[0].boxes)
        if k in (0, 2, 4):
            want = oracle.single_dbscan(variant, X, Y, eps, m, cut)["labels"]
            assert np.array_equal(rs[2].labels, want)
    for h in hs:
        h.close()


@pytest.mark.parametrize("variant", ["v2", "v1"])
def test_layouts_from_the_fine_layout(variant):
    """cl_set_eps_list: the eps values share a divisor -- every layout is merged from ONE sort by strips of that width (2, 3, 4 and
    1 runs per strip, revisits, a value outside the list in between), and the labels equal a handle that sorts every layout and
    the oracle"""
    X, Y = synth_chrom(400000, 9000000, 21)
    a = api.Chromosome(X, Y)
    b = api.Chromosome(X, Y)
    a.set_sort_index(1)
    b.set_sort_index(1)
    a.set_eps_list([1500, 3000, 4500, 6000])
    runs = [(3000, 8, 0), (3000, 6, 2500), (4500, 8, 2700), (6000, 10, 0), (1500, 5, 0), (2000, 6, 1000), (4500, 6, 0), (3000, 8, 3100)]
    for k, (eps, m, cut) in enumerate(runs):
        ra = a.cluster(variant, eps, m, cut)
        rb = b.cluster(variant, eps, m, cut)
        assert np.array_equal(ra.labels, rb.labels), (variant, eps, m, cut)
        assert ra.n_clusters == rb.n_clusters and np.array_equal(ra.boxes, rb.boxes)
        if k in (1, 2, 3):
            want = oracle.single_dbscan(variant, X, Y, eps, m, cut)["labels"]
            assert np.array_equal(ra.labels, want), (variant, eps, m, cut)
    # a list without a useful divisor, and more runs per strip than the merge takes: plain sorts, same results
    for lst in ([1000, 1007], [500, 6000]):
        a.set_eps_list(lst)
        for eps in lst:
            ra, rb = a.cluster(variant, eps, 6, 900), b.cluster(variant, eps, 6, 900)
            assert np.array_equal(ra.labels, rb.labels), (variant, lst, eps)
    a.set_eps_list([])
    a.close()
    b.close()


def test_fine_layout_on_chr21_edges():
    """real data (ragged strips, empty strips, pile-ups at the chromosome's ends) through the merged layouts"""
    X, Y = G.chr21_xy()
    a = api.Chromosome(X, Y)
    a.set_sort_index(1)
    a.set_eps_list([750, 1500, 2250])
    for eps, m, cut in [(750, 5, 0), (1500, 5, 4601), (2250, 4, 0), (1500, 3, 0)]:
        got = a.cluster("v2", eps, m, cut).labels
        want = oracle.single_dbscan("v2", X, Y, eps, m, cut)["labels"]
        assert np.array_equal(got, want), (eps, m, cut)
    a.close()


def test_sweep_plan_and_drop_indexes_leave_results_alone():
    """cl_sweep_plan announces both lists of a sweep in one call (cLoops/pipe.py:247-248); cl_chrom_drop_indexes forgets every order and
    count the handle has derived (what bench.py's cold_sweep_s pays for again): the labels of every run equal a plain handle's, before
    and after the drop, and the plan makes the later runs of an eps re-use its words (region mode 2)"""
    from cloops_amd.synth import synth_chrom
    X, Y = synth_chrom(120000, 248956422 // 60, 17)
    a = api.Chromosome(X, Y)
    b = api.Chromosome(X, Y)
    try:
        b.set_count_reuse(False)
        eps_list, mps = [3000, 4500, 6000], [30, 20]
        for rnd in range(2):
            a.sweep_plan(eps_list, mps)
            cut = 0
            for ep in eps_list:
                for k, m in enumerate(mps):
                    ra = a.cluster("v2", ep, m, cut)
                    assert a.last_region_mode() == (0 if k == 0 else 2), (rnd, ep, m)
                    rb = b.cluster("v2", ep, m, cut)
                    assert np.array_equal(ra.labels, rb.labels), (rnd, ep, m, cut)
                    cut += 350
            a.sweep_plan([], [])
            a.drop_indexes()                                  # the second round sorts everything again
        with pytest.raises(Exception):
            a.sweep_plan([3000, -1], [30])
    finally:
        a.close(); b.close()
