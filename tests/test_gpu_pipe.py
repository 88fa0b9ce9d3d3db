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
