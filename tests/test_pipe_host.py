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
