"""GPU: the reference-shaped surface (classes with `.labels`, singleDBSCAN / runDBSCAN /
runSweep) end to end on the real HIP path, against reference-made golden vectors."""
import numpy as np
import pytest

import golden_util as G
import pipe_checks
from cloops_amd import pipe, _lib
from cloops_amd.cDBSCAN import cDBSCAN as cDBSCAN1
from cloops_amd.cDBSCAN2 import cDBSCAN as cDBSCAN2
from cloops_amd.blockDBSCAN import blockDBSCAN

pytestmark = pytest.mark.gpu

CLS = {"v1": cDBSCAN1, "v2": cDBSCAN2, "block": blockDBSCAN}


@pytest.fixture()
def jd(tmp_path):
    pipe.CACHE.clear()
    yield pipe_checks.write_chr21_jd(tmp_path)
    pipe.CACHE.clear()


def test_run_dbscan_chain_gpu(jd):
    pipe_checks.check_run_dbscan_chain(pipe, jd)


def test_sweep_gpu(jd):
    pipe_checks.check_sweep(pipe, jd)


def test_sweep_fast_gpu_statistics(jd):
    pipe_checks.check_sweep_fast(pipe, jd)


def test_dispatch_under_variant1(jd, monkeypatch):
    """scripts/jd2saturation:56-127 and scripts/callStripes:37-72 run the same dispatch with `cDBSCAN` (v1) imported as
    DBSCAN: runDBSCAN step by step (boxes in order, distance multisets, both stderr lines, the cut), runSweep and
    runSweepFast, against the goldens the reference's own dispatch functions produced under variant 1"""
    monkeypatch.setattr(pipe, "DBSCAN_VARIANT", "v1")
    pipe_checks.check_run_dbscan_chain(pipe, jd, variant="v1")
    pipe_checks.check_sweep(pipe, jd, variant="v1")
    pipe_checks.check_sweep_fast(pipe, jd, variant="v1")


@pytest.mark.parametrize("variant", ["v2", "v1", "block"])
@pytest.mark.parametrize("device_labels", [True, False])
def test_dist_stats_match_numpy(jd, variant, device_labels):
    """K7 reductions against numpy on the lists the reference would build (pipe.py:63,106-109): one-pass summary
    (counts, sum / sum of squares of the shifted log2 distances, log-binned histogram) and the exact median through
    the bin refinement -- from the sorted arrays (rotated variants, with and without row-order labels) and from
    row-order labels (block variant)."""
    import oracle
    import golden_util as G
    from cloops_amd import api, ests
    X, Y = G.chr21_xy()
    ch = api.Chromosome(X, Y)
    ch.set_device_labels(device_labels)
    for eps, minPts, cut in ((500, 5, 0), (1000, 5, 4601), (1000, 4, 300)):
        ch.cluster(variant, eps, minPts, cut, want_labels=False)
        ref = oracle.single_dbscan(variant, X, Y, eps, minPts, cut)
        st = ch.dist_summary(cut)
        assert st["n_all"] == [len(ref["dis"]), len(ref["dss"])]
        for g, arr in ((0, ref["dis"]), (1, ref["dss"])):
            a = np.abs(arr)
            a = a[a > 0]
            assert st["n_pos"][g] == len(a)
            sumlog = st["sumx"][g] + st["xshift"] * len(a)
            assert abs(sumlog - np.log2(a).sum()) < 1e-6 * max(1.0, np.log2(a).sum())
            sq = st["sumxx"][g] - st["sumx"][g] ** 2 / len(a)
            assert abs(np.sqrt(sq / len(a)) - np.log2(a).std()) < 1e-9
        a = np.abs(ref["dss"]); a = a[a > 0]
        srt = np.sort(a).astype(np.int64)
        assert np.array_equal(st["loghist"], np.bincount([ests.logbin(d) for d in srt.tolist()], minlength=3840))
        n1 = len(srt)
        class R(object):
            chrom = ch
        got = pipe._select_kth([R], cut, st["loghist"], sorted({(n1 - 1) // 2, n1 // 2}))
        assert got[0] == srt[(n1 - 1) // 2] and got[-1] == srt[n1 // 2]
        # any rank, also in wide bins (two refinement passes)
        for rk in (0, n1 // 7, n1 - 1):
            assert pipe._select_kth([R], cut, st["loghist"], [rk])[0] == srt[rk]
    ch.close()


@pytest.mark.parametrize("variant", ["v1", "v2", "block"])
def test_class_labels_dict(variant):
    """`DBSCAN(mat, eps, minPts).labels` == the reference's dict (non-contiguous ids)."""
    for k, ids, X, Y, eps, minPts, gold in list(G.family_cases("adversarial"))[:25]:
        mat = np.stack([ids, X, Y], 1)
        db = CLS[variant](mat, eps, minPts)
        want = {int(i): int(c) for i, c in zip(ids, gold[variant]) if c >= 0}
        assert db.labels == want
        assert db.eps == eps and db.minPts == minPts and db.cw == eps


def test_error_behaviour():
    empty = np.zeros((0, 3), np.int64)
    assert cDBSCAN2(empty, 100, 5).labels == {}                    # cDBSCAN2 on empty -> {}
    for cls in (cDBSCAN1, blockDBSCAN):
        with pytest.raises(IndexError):                            # cDBSCAN.py:77 / blockDBSCAN.py:74
            cls(empty, 100, 5)
    mat = np.array([[0, 10, 20], [1, 11, 21]])
    for cls in (cDBSCAN1, cDBSCAN2, blockDBSCAN):
        with pytest.raises(ZeroDivisionError):
            cls(mat, 0, 5)
    with pytest.raises(_lib.CloopsHipError) as ei:                 # documented deviation: X > Y for cDBSCAN2
        cDBSCAN2(np.array([[0, 30, 20], [1, 11, 21]]), 5, 2)
    assert ei.value.code == _lib.CL_ERR_DOMAIN


def test_lists_and_mutation_semantics():
    """mat may be a list of lists (rows are only indexed d[0..2]) and is never mutated."""
    rows = [[7, 100, 200], [9, 101, 201], [11, 102, 202], [13, 5000, 9000]]
    keep = [list(r) for r in rows]
    db = cDBSCAN2(rows, 10, 3)
    assert rows == keep
    assert db.labels == {7: 0, 9: 0, 11: 0}


def test_end_to_end_bedpe_to_loop_file(tmp_path):
    """config 1 end to end on the GPU: BEDPE text -> .jd -> chained sweep -> significance -> `.loop`,
    text-identical to the golden table of the (converted) reference pipeline."""
    import gzip
    import os
    X, Y = G.chr21_xy()
    bed = os.path.join(str(tmp_path), "in.bedpe.gz")
    with gzip.open(bed, "wt") as fh:                       # a BEDPE whose mid-points are exactly (X, Y)
        for x, y in zip(X.tolist(), Y.tolist()):
            fh.write("chr21\t%d\t%d\tchr21\t%d\t%d\tid\t1\t+\t-\n" % (x, x, y, y))
    fout = os.path.join(str(tmp_path), "run")
    pipe.CACHE.clear()
    steps = pipe.pipe([bed], fout, [500, 1000, 2000], [5], tmp=0, hic=0)
    assert [s.get("cut_out") for s in steps] == [4601, 13532, 11103]
    got = open(fout + ".loop").read()
    want = open(os.path.join(G.GOLD, "chr21_v2.loop")).read()
    assert got == want
    assert not os.path.isdir(fout)                         # pipe.py:294-295 removes the working directory


def test_end_to_end_auto_eps(tmp_path):
    """`-eps 0` (cLoops/pipe.py:231-239): eps = 2 x the fragment size estimated from the PETs mapped to different strands
    (io.py:62-129 also drops duplicate PETs); the run then equals a run with that eps given on the de-duplicated file."""
    import gzip
    import os
    from cloops_amd import io as cio, ests
    X, Y = G.chr21_xy()
    bed = os.path.join(str(tmp_path), "in.bedpe.gz")
    with gzip.open(bed, "wt") as fh:                       # mid-points (X, Y); two of three PETs on different strands; 500 duplicates
        rows = list(zip(X.tolist(), Y.tolist()))
        for k, (x, y) in enumerate(rows + rows[:500]):
            fh.write("chr21\t%d\t%d\tchr21\t%d\t%d\tid\t1\t+\t%s\n" % (x, x, y, y, "-" if k % 3 else "+"))
    d = os.path.join(str(tmp_path), "p")
    os.mkdir(d)
    cfs, ds = cio.parseRawBedpe([bed], d, [], 0)
    eps = ests.estFragSize(ds) * 2
    assert eps > 0 and len(ds) > 1000
    uniq = os.path.join(str(tmp_path), "uniq.bedpe.gz")     # the same PETs without the duplicates, for the run with eps given
    import joblib
    m = joblib.load(cfs[0])
    with gzip.open(uniq, "wt") as fh:
        for _, x, y in m.tolist():
            fh.write("chr21\t%d\t%d\tchr21\t%d\t%d\tid\t1\t+\t-\n" % (x, x, y, y))
    pipe.CACHE.clear()
    a = pipe.pipe([bed], os.path.join(str(tmp_path), "auto"), 0, [5], tmp=0, hic=0)
    pipe.CACHE.clear()
    b = pipe.pipe([uniq], os.path.join(str(tmp_path), "given"), [eps], [5], tmp=0, hic=0)
    pipe.CACHE.clear()
    strip = lambda steps: [{k: v for k, v in st.items() if k != "wall_s"} for st in steps]      # (wall_s: the step's wall time)
    assert [s["eps"] for s in a] == [eps] and strip(a) == strip(b)
    assert open(os.path.join(str(tmp_path), "auto.loop")).read() == open(os.path.join(str(tmp_path), "given.loop")).read()


def test_sig_counts_kernel_vs_sets():
    """K8 interval counts == the set-based host restatement (cModel.CoverageModel), incl. a cut filter"""
    import fake_backend
    import pipe_checks
    from cloops_amd import api, cModel
    X, Y = G.chr21_xy()
    z, meta = pipe_checks.pipe_golden()
    recs = [["chr21", int(a), int(b), "chr21", int(c), int(d)] for a, b, c, d in z["v2_filtered"][:200]]
    _, _, _, wins = cModel._windows(recs)
    ch = api.Chromosome(X, Y)
    for cut in (0, 4601):
        got, n = ch.sig_counts(wins, cut)
        want, n2 = fake_backend.FakeChromosome(X, Y).sig_counts(wins, cut)
        assert n == n2
        assert np.array_equal(got, want)
    ch.close()


@pytest.mark.parametrize("flip", [False, True])
def test_dist_stats_cut_sources(jd, flip):
    """the PETs below the cut come from the upload's distance histogram (cut < 65536, no row with Y < X) or from a pass
    over the rows (rows with Y < X; a cut beyond the histogram): both against numpy on the reference's lists"""
    import oracle
    import golden_util as G
    from cloops_amd import api
    X, Y = G.chr21_xy()
    if flip:                                             # some PETs stored with X > Y: pipe.py:63 keeps them (d < cut), ests.py:42 takes |d|
        X, Y = X.copy(), Y.copy()
        sel = np.arange(len(X)) % 17 == 0
        X[sel], Y[sel] = Y[sel].copy(), X[sel].copy()
    ch = api.Chromosome(X, Y)
    ch.set_device_labels(False)
    for eps, minPts, cut in ((1000, 5, 4601), (1000, 5, 70000), (500, 4, 65535), (500, 4, 65536)):
        ch.cluster("v1", eps, minPts, cut, want_labels=False)
        ref = oracle.single_dbscan("v1", X, Y, eps, minPts, cut)
        st = ch.dist_summary(cut)
        assert st["n_all"] == [len(ref["dis"]), len(ref["dss"])], (flip, cut)
        for g, arr in ((0, ref["dis"]), (1, ref["dss"])):
            a = np.abs(arr)
            a = a[a > 0]
            assert st["n_pos"][g] == len(a)
            if len(a):
                sumlog = st["sumx"][g] + st["xshift"] * len(a)
                assert abs(sumlog - np.log2(a).sum()) < 1e-6 * max(1.0, np.log2(a).sum())
    ch.close()


def test_two_workers_on_one_device(monkeypatch):
    """the multi-device dispatch of runDBSCAN / runSweepFast (cLoops/pipe.py:117: one worker per chromosome group) with what one GPU can
    exercise of it: CLOOPS_DEVICES=0,0 = two workers (host threads, LPT shares of the chromosomes) on device 0 -- same result
    as the single-worker form.  (RCCL refuses two ranks on one device: world > 1 stays untested until a multi-GPU node exists.)"""
    from cloops_amd.synth import synth_chrom
    chroms = {}
    for k, (n, L) in enumerate(((60000, 40000000), (45000, 30000000), (30000, 20000000), (20000, 15000000))):
        chroms["chr%d-chr%d" % (k + 1, k + 1)] = synth_chrom(n, L, 50 + k)

    def run_all():
        pipe.CACHE.clear()
        fs = [pipe.CACHE.put_arrays(name, X, Y) for name, (X, Y) in chroms.items()]
        one = pipe.runDBSCAN(fs, 2000, 5, cut=0)
        sweep = pipe.runSweepFast(fs, [1000, 2000], [6, 4], cut=0)
        pipe.CACHE.clear()
        return one, sweep

    monkeypatch.delenv("CLOOPS_DEVICES", raising=False)
    (dI1, dS1, dis1, dss1), (sI1, cut1, cuts1, steps1) = run_all()
    monkeypatch.setenv("CLOOPS_DEVICES", "0,0")
    assert pipe._devices() == [0, 0]
    (dI2, dS2, dis2, dss2), (sI2, cut2, cuts2, steps2) = run_all()
    assert list(dI1) == list(dI2) and all(dI1[k]["records"] == dI2[k]["records"] for k in dI1)
    assert dS1 == dS2 and dis1 == dis2 and dss1 == dss2
    assert cut1 == cut2 and cuts1 == cuts2
    assert [(s["n_inter"], s["n_self"], s.get("cut_out")) for s in steps1] == [(s["n_inter"], s["n_self"], s.get("cut_out")) for s in steps2]
    assert list(sI1) == list(sI2) and all(np.array_equal(sI1[k]["boxes"], sI2[k]["boxes"]) for k in sI1)


def test_handle_moves_between_shared_streams():
    """cl_chrom_set_stream: an idle handle made on one shared stream clusters the same on another; a handle with a stream of its own
    refuses; the sweep driver's LPT re-deal (pipe.STREAMS.rebalance) leaves the results of a sweep untouched"""
    from cloops_amd import api, pipe, _lib
    from cloops_amd.synth import synth_chrom
    lib = _lib.load()
    s1, s2 = lib.cl_stream_create(0), lib.cl_stream_create(0)
    X, Y = synth_chrom(120000, 3000000, 4)
    a = api.Chromosome(X, Y, stream=s1)
    want = a.cluster("v2", 2000, 5, 0).labels.copy()
    a.set_stream(s2)
    assert np.array_equal(a.cluster("v2", 2000, 5, 300).labels, api.Chromosome(X, Y).cluster("v2", 2000, 5, 300).labels)
    a.set_stream(s1)
    assert np.array_equal(a.cluster("v2", 2000, 5, 0).labels, want)
    own = api.Chromosome(X, Y)
    with pytest.raises(Exception):
        own.set_stream(s1)
    a.close()
    own.close()
    # residents uploaded smallest first land unevenly on the shared streams; the sweep deals them again and gives the same chain
    parts = [synth_chrom(n, 4000000, 50 + k) for k, n in enumerate((20000, 30000, 50000, 90000, 160000))]
    fs = [pipe.CACHE.put_arrays("rb%d-rb%d" % (k, k), x, y) for k, (x, y) in enumerate(parts)]
    try:
        d1, cut1, cuts1, _ = pipe.runSweepFast(fs, [2000, 3000], [8, 5], cut=0)
        keep = pipe.STREAMS.rebalance
        pipe.STREAMS.rebalance = lambda chroms: None
        try:
            d2, cut2, cuts2, _ = pipe.runSweepFast(fs, [2000, 3000], [8, 5], cut=0)
        finally:
            pipe.STREAMS.rebalance = keep
        assert cut1 == cut2 and cuts1 == cuts2 and list(d1) == list(d2)
        for k in d1:
            assert np.array_equal(d1[k]["boxes"], d2[k]["boxes"])
    finally:
        for f in fs:
            pipe.CACHE.drop(f)
