import os, sys, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import pipe, cModel
from cloops_amd.synth import synth_genome
fs = [pipe.CACHE.put_arrays("%s-%s" % (n, n), X, Y) for n, X, Y in synth_genome(5000000, cfg=1)]
eps, minPts, hic = pipe.MODES[1]
dataI, cut, cuts, steps = pipe.runSweepFast(fs, eps, minPts, cut=0)
records = {key: {"f": v["f"], "records": pipe._records(key, v["boxes"])} for key, v in dataI.items()}
with tempfile.TemporaryDirectory() as td:
    pr = cProfile.Profile(); pr.enable()
    cModel.runStat(records, minPts, 0, 1, os.path.join(td, "o"), hic)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
