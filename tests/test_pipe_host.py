"""CPU: host logic of cloops_amd.pipe (dispatch, record classification, report lines, sweep
chain, combine/filter) against the reference-made golden vectors, with the GPU replaced by
the oracle-backed FakeChromosome (tests/fake_backend.py)."""
import numpy as np
import pytest

import fake_backend
import pipe_checks
import refload
from cloops_amd import api, pipe


@pytest.fixture()
def cpu_pipe(monkeypatch, tmp_path):
    monkeypatch.setattr(api, "Chromosome", fake_backend.FakeChromosome)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    f = pipe_checks.write_chr21_jd(tmp_path)
    yield pipe, f
    pipe.CACHE.clear()


def test_run_dbscan_matches_reference_outputs(cpu_pipe):
    p, f = cpu_pipe
    pipe_checks.check_run_dbscan_chain(p, f)


def test_sweep_chain_combine_filter(cpu_pipe):
    p, f = cpu_pipe
    pipe_checks.check_sweep(p, f)


def test_single_dbscan_all_filtered(cpu_pipe):
    p, f = cpu_pipe
    key, ff, dataI, dataS, dis, dss = p.singleDBSCAN(f, 500, 5, cut=10 ** 9)     # pipe.py:64-65
    assert key == ("chr21", "chr21") and dataI == [] and dataS == [] and dis == []
    assert len(dss) == 99674


def test_filter_uses_floor_division():
    data = {("c", "c"): {"f": "x", "records": [["c", 0, 3, "c", 10, 13], ["c", 0, 3, "c", 10, 12]]}}
    out = pipe.filterClusterByDis(data, 10)          # (10+13)//2 - (0+3)//2 = 10 ; (10+12)//2 - 1 = 10
    assert len(out[("c", "c")]["records"]) == 2
    out = pipe.filterClusterByDis(data, 11)
    assert len(out[("c", "c")]["records"]) == 0


@pytest.mark.skipif(not refload.available(), reason="reference checkout not present")
def test_ests_matches_reference_function():
    from cloops_amd.ests import estIntSelCutFrag
    ref = refload.ref_ests().estIntSelCutFrag
    rng = np.random.default_rng(3)
    for _ in range(20):
        di = np.exp(rng.normal(10, 1.5, 3000)).astype(np.int64).astype(float)
        ds = np.exp(rng.normal(6, 1.0, 5000)).astype(np.int64).astype(float)
        assert estIntSelCutFrag(di, ds) == ref(di, ds)


def test_combine_steps_equals_sequential_combine_twice():
    """the array form used by runSweepFast == repeated combineTwice (pipe.py:155-174), order included"""
    rng = np.random.default_rng(4)
    pool = rng.integers(0, 50, (60, 4)).astype(np.int64)
    steps = []
    for k in range(6):
        idx = rng.integers(0, len(pool), int(rng.integers(0, 30)))
        b = pool[idx]
        if k == 2 and len(b) > 3:
            b = np.concatenate([b, b[:2]])               # duplicates inside one step are all kept
        steps.append(b)
    dataI = {}
    for b in steps:
        if len(b) == 0:
            continue
        d2 = {("c", "c"): {"f": "x", "records": [["c", int(r[0]), int(r[1]), "c", int(r[2]), int(r[3])] for r in b]}}
        dataI = pipe.combineTwice(dataI, d2)
    want = np.asarray([[r[1], r[2], r[4], r[5]] for r in dataI[("c", "c")]["records"]], dtype=np.int64)
    assert np.array_equal(pipe._combine_steps(steps), want)
    # truncated hashes: different boxes share a sort key, the exact repair of those runs must kick in
    for bits in (1, 2, 3, 5):
        assert np.array_equal(pipe._combine_steps(steps, _hash_bits=bits), want)


def test_sweep_fast_host_logic(cpu_pipe):
    """runSweepFast (statistics instead of lists) against the reference-made chain, CPU backend"""
    p, f = cpu_pipe
    pipe_checks.check_sweep_fast(p, f)


def test_filter_candidate_stripes_semantics():
    """scripts/callStripes:75-86: keep records with >= pets PETs whose side lengths differ by more than
    lengthFoldDiff (py2 integer division)"""
    from cloops_amd import stripes
    rs = {("c", "c"): [["c", 0, 1000, "c", 0, 10, 300],       # 100x longer in X: kept
                       ["c", 0, 10, "c", 0, 1000, 300],       # 100x longer in Y: kept
                       ["c", 0, 100, "c", 0, 100, 300],       # square: dropped
                       ["c", 0, 1000, "c", 0, 10, 100],       # too few PETs: dropped
                       ["c", 0, 209, "c", 0, 10, 300]]}       # 209 // 10 = 20, not > 20: dropped
    out = stripes.filterCandidateStripes(rs, pets=200, lengthFoldDiff=20)
    assert out[("c", "c")] == [["c", 0, 1000, "c", 0, 10, 300], ["c", 0, 10, "c", 0, 1000, 300]]


# ---- resident-chromosome cache: a sweep keeps every handle it uses alive ---------------------------
class _ClosableFake(fake_backend.FakeChromosome):
    """like the real handle: close() destroys it, any later call fails"""
    closed_calls = 0

    def close(self):
        self.dead = True

    def cluster_async(self, *a, **kw):
        if getattr(self, "dead", False):
            type(self).closed_calls += 1
            raise RuntimeError("null chromosome handle")
        return fake_backend.FakeChromosome.cluster_async(self, *a, **kw)


def _write_many_jd(tmp_path, n_files, n=400):
    import joblib
    from cloops_amd.synth import synth_chrom
    fs = []
    for k in range(n_files):
        X, Y = synth_chrom(n, 200000, 900 + k)
        f = str(tmp_path / ("c%d-c%d.jd" % (k, k)))
        joblib.dump(np.stack([np.arange(n), X, Y], 1).astype(np.int64), f)
        fs.append(f)
    return fs


def test_sweep_over_more_chromosomes_than_the_cache_keeps(monkeypatch, tmp_path):
    """70 per-chromosome files > max_items = 64 (a scaffold-level assembly): no resident that the sweep
    uses may be evicted / closed while the sweep runs; afterwards the cache shrinks back"""
    monkeypatch.setattr(api, "Chromosome", _ClosableFake)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    fs = _write_many_jd(tmp_path, 70)
    _ClosableFake.closed_calls = 0
    dataI, cut, cuts, steps = pipe.runSweepFast(fs, [800], [4, 3], cut=0)
    assert _ClosableFake.closed_calls == 0
    assert len(steps) == 2 and steps[0]["n_in"] == 70 * 400
    assert len(pipe.CACHE._items) <= pipe.CACHE.max_items
    assert all(r.pins == 0 for r in pipe.CACHE._items.values())
    pipe.CACHE.clear()


def test_failed_enqueue_releases_every_lock(monkeypatch, tmp_path):
    """cluster_async raising for chromosome k must not leave chromosomes 0..k-1 locked / in flight"""
    class Failing(fake_backend.FakeChromosome):
        def cluster_async(self, variant, eps, minPts, cut=0, want_labels=True, want_boxes=True):
            if self.n == 403:
                raise RuntimeError("boom")
            return fake_backend.FakeChromosome.cluster_async(self, variant, eps, minPts, cut, want_labels, want_boxes)
    monkeypatch.setattr(api, "Chromosome", Failing)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    from cloops_amd.synth import synth_chrom
    fs = [pipe.CACHE.put_arrays("k%d-k%d" % (k, k), *synth_chrom(400 + k, 200000, 70 + k)) for k in range(5)]
    with pytest.raises(RuntimeError):
        pipe.runSweepFast(fs, [800], [4], cut=0)
    for f in fs:
        r = pipe.CACHE.get(f)
        assert not r.lock.locked() and r.pins == 0
        assert not getattr(r.chrom, "_pending", [])
    pipe.CACHE.clear()


def test_cache_get_without_device_preference_keeps_the_resident(monkeypatch, tmp_path):
    """the significance stage asks for a chromosome without a device preference: the resident copy on
    another GPU is reused, not reloaded onto GPU 0"""
    monkeypatch.setattr(api, "Chromosome", fake_backend.FakeChromosome)
    monkeypatch.setattr(api, "device_count", lambda: 2)
    pipe.CACHE.clear()
    f = _write_many_jd(tmp_path, 1)[0]
    r1 = pipe.CACHE.get(f, 1)
    assert r1.device == 1 and pipe.CACHE.get(f) is r1 and pipe.CACHE.get(f, None) is r1
    assert pipe.CACHE.get(f, 0) is not r1
    pipe.CACHE.clear()


def test_cut_on_an_integer_boundary_is_settled_from_the_lists(cpu_pipe, monkeypatch):
    """rcut = int(2 ** cut) truncates: when the statistics-based 2**cut lands within rounding noise of an integer
    the step is re-derived from the distance lists like the reference does -- forced here by a huge margin;
    the chain must not change and the steps say so"""
    p, f = cpu_pipe
    z, meta = pipe_checks.pipe_golden()
    monkeypatch.setattr(p, "CUT_RECHECK_MARGIN", 10.0)
    dataI, cut, cuts, steps = p.runSweepFast([f], [500, 1000, 2000], [5], cut=0)
    assert all(s.get("cut_rechecked") for s in steps)
    assert [s.get("cut_out") for s in steps] == [s.get("cut_out") for s in meta["v2"]["steps"]]
    assert [s.get("frags") for s in steps] == [s.get("frags") for s in meta["v2"]["steps"]]
    assert np.array_equal(dataI[("chr21", "chr21")]["boxes"], z["v2_filtered"])


def test_sweep_fast_key_order_is_first_appearance(monkeypatch):
    """combineTwice (cLoops/pipe.py:155-174) inserts a chromosome's key when it FIRST yields an inter-ligation box;
    runStat walks the dict, so the key order fixes the row order of the .loop file.  A chromosome that comes first in
    file order but gets its first box only in the second step must come second -- in runSweep and runSweepFast alike."""
    monkeypatch.setattr(api, "Chromosome", fake_backend.FakeChromosome)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    rng = np.random.RandomState(5)

    def blob(x, y, k):
        return x + rng.randint(0, 300, k), y + rng.randint(0, 300, k)

    ax, ay = blob(100000, 600000, 5)                      # chrA: one inter-ligation cluster of 5 PETs (only minPts 4 sees it)
    sx = 300000 + rng.randint(0, 300, 12)
    bx, by = [np.concatenate(t) for t in zip(blob(150000, 700000, 12), (sx, sx + rng.randint(20, 200, 12)))]   # chrB: inter + self, 12 each
    fa = pipe.CACHE.put_arrays("chrA-chrA", ax, ay)
    fb = pipe.CACHE.put_arrays("chrB-chrB", bx, by)
    try:
        slow = pipe.runSweep([fa, fb], [1000], [8, 4], cut=0)
        fast = pipe.runSweepFast([fa, fb], [1000], [8, 4], cut=0)
        assert [s["n_inter"] for s in slow[3]] == [1, 2] and slow[1] > 0
        assert list(slow[0]) == [("chrB", "chrB"), ("chrA", "chrA")]
        assert list(fast[0]) == list(slow[0])
        assert fast[1] == slow[1] and fast[2] == slow[2]
    finally:
        pipe.CACHE.clear()


def test_two_sweeps_over_the_same_residents_do_not_interleave(monkeypatch, tmp_path):
    """the candidate buffer / layout of a handle belong to one sweep at a time: two threads sweeping the same files get
    the same result as one after the other"""
    import threading
    monkeypatch.setattr(api, "Chromosome", fake_backend.FakeChromosome)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    fs = _write_many_jd(tmp_path, 3, n=600)
    try:
        want = pipe.runSweepFast(fs, [800, 1200], [4, 3], cut=0)
        got = [None, None]

        def run(k):
            got[k] = pipe.runSweepFast(fs, [800, 1200], [4, 3], cut=0)
        ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for g in got:
            assert g[1] == want[1] and g[2] == want[2] and g[0].keys() == want[0].keys()
            for k in g[0]:
                assert np.array_equal(g[0][k]["boxes"], want[0][k]["boxes"])
    finally:
        pipe.CACHE.clear()


def test_replaced_resident_is_closed_by_its_last_user(monkeypatch, tmp_path):
    monkeypatch.setattr(api, "Chromosome", _ClosableFake)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    X = np.arange(10, dtype=np.int64) * 100
    f = pipe.CACHE.put_arrays("c-c", X, X + 50)
    with pipe.CACHE.pinned([f]) as rs:
        old = rs[0]
        pipe.CACHE.put_arrays("c-c", X, X + 60)          # replaces the resident a sweep still holds
        assert not getattr(old.chrom, "dead", False) and old.replaced
    assert getattr(old.chrom, "dead", False)              # closed when the pin dropped
    pipe.CACHE.clear()


def test_forced_cuts_replace_the_estimated_chain(cpu_pipe):
    """runSweepFast(forced_cuts=...) (bench.py's scaling proxy: one rank's share replays the genome-wide chain): every step
    still runs and estimates, but hands the given cut to the next step"""
    p, f = cpu_pipe
    free = p.runSweepFast([f], [500, 1000], [5], cut=0)
    forced = p.runSweepFast([f], [500, 1000], [5], cut=0, forced_cuts=[3000, 7000])
    assert [s["cut_in"] for s in forced[3]] == [0, 3000] and [s["cut_out"] for s in forced[3]] == [3000, 7000]
    assert forced[2] == [0, 3000, 7000] and forced[1] == 3000
    assert [s["cut_out"] for s in free[3]] != [3000, 7000]
    assert forced[3][0]["n_inter"] == free[3][0]["n_inter"]          # the first step is the same work
    none = p.runSweepFast([f], [500, 1000], [5], cut=0, forced_cuts=[None, None])
    assert none[2] == free[2]


def test_device_tables_are_consumed_under_the_sweeps_locks(monkeypatch, tmp_path):
    """runSweepFast(finish_device=True, device_consumer=...): the consumer sees every chromosome's (pointer, rows) WHILE the residents are
    pinned in the cache and their sweep locks are held -- nothing can evict, free or re-sweep a handle under the gather
    (cLoops/pipe.py:119-127: the parent merges its workers' results; here the merge reads device memory of live handles)"""
    monkeypatch.setattr(api, "Chromosome", fake_backend.FakeChromosome)
    monkeypatch.setattr(api, "device_count", lambda: 1)
    pipe.CACHE.clear()
    fs = _write_many_jd(tmp_path, 3, n=600)
    seen = {}

    def consumer(dataI):
        res = [pipe.CACHE.get(v["f"]) for v in dataI.values()]
        seen["locked"] = all(r.sweep_lock.locked() for r in res)
        seen["pinned"] = all(r.pins > 0 for r in res)
        seen["rows"] = {k: (v["dev_rows"], v["n_rows"]) for k, v in dataI.items()}
        return sum(v["n_rows"] for v in dataI.values())
    try:
        want = pipe.runSweepFast(fs, [800, 1200], [4, 3], cut=0)
        got = pipe.runSweepFast(fs, [800, 1200], [4, 3], cut=0, finish_device=True, device_consumer=consumer)
        assert seen["locked"] and seen["pinned"]
        assert got[0].gathered == sum(len(v["boxes"]) for v in want[0].values())
        assert {k: n for k, (_, n) in seen["rows"].items()} == {k: len(v["boxes"]) for k, v in want[0].items()}
        assert got[1:3] == want[1:3]
        assert not any(pipe.CACHE.get(f).sweep_lock.locked() for f in fs)
    finally:
        pipe.CACHE.clear()
