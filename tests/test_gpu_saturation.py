"""GPU: the re-sampling flow of scripts/jd2saturation on the real HIP path -- samples gathered on the device
(cl_chrom_subsample), variant 1 at minPts 1 .. 5 with and without a cut, against golden vectors made by the script's own
functions wired to the real cDBSCAN class; the whole flow BEDPE-free from a resident chromosome to the re-sampling table."""
import os

import numpy as np
import pytest

import golden_util as G
import oracle
import saturation_checks
from cloops_amd import api, pipe, saturation, _lib

pytestmark = pytest.mark.gpu


def test_flow_against_the_scripts_own_functions():
    saturation_checks.check_flow(pipe, saturation)


def test_subsample_is_a_device_gather():
    X, Y = G.chr21_xy()
    rng = np.random.default_rng(5)
    rows = rng.permutation(len(X))[:30000]
    rows[:100] = rows[100:200]                               # repeated rows are rows too
    src = api.Chromosome(X, Y)
    sub = src.subsample(rows)
    ref = api.Chromosome(X[rows], Y[rows])
    try:
        for variant, eps, m, cut in (("v1", 1000, 3, 0), ("v2", 2000, 5, 4601), ("block", 500, 4, 0)):
            a, b = sub.cluster(variant, eps, m, cut), ref.cluster(variant, eps, m, cut)
            assert np.array_equal(a.labels, b.labels) and np.array_equal(a.boxes, b.boxes)
        want = oracle.single_dbscan("v1", X[rows], Y[rows], 1000, 3, 0)["labels"]
        assert np.array_equal(sub.cluster("v1", 1000, 3).labels, want)
        assert src.cluster("v1", 1000, 3).n_clusters > 0     # the source is untouched
        with pytest.raises(_lib.CloopsHipError):
            src.subsample(np.asarray([0, len(X)]))
        empty = src.subsample(np.zeros(0, np.int64))
        assert empty.n == 0
        empty.close()
    finally:
        src.close(); sub.close(); ref.close()


def test_whole_flow_writes_the_resampling_table(tmp_path):
    X, Y = G.chr21_xy()
    pipe.CACHE.clear()
    jd = pipe.CACHE.put_arrays("chr21-chr21", X, Y)
    fout = os.path.join(str(tmp_path), "sat")
    try:
        np.random.seed(3)
        ds = saturation.jd2saturation(jd, fout, [1000, 2000], 5, 1, 2, hic=0, cut=0)
        assert os.path.isfile(os.path.join(fout, "sat.loop")) and os.path.isfile(os.path.join(fout, "depth_0.5_rep_0.loop"))
        assert os.path.isfile(os.path.join(fout, "sat_ResamplingRatios.txt"))
        assert ds.shape == (1, 1) and 0.0 <= float(ds.iloc[0, 0]) <= 100.0
        assert saturation.jd2saturation(jd, fout, [1000, 2000], 5, 1, 2) is None      # the working directory exists: return (:231-234)
    finally:
        pipe.CACHE.clear()
