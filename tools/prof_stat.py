"""cProfile of the significance stage (cModel.runStat) behind a sweep (developer tool; keep n small)."""
import os, sys, tempfile, time, cProfile, pstats, faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("DBG_T", "100")), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import pipe, cModel
from cloops_amd.synth import synth_genome
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000000
fs = [pipe.CACHE.put_arrays("%s-%s" % (nm, nm), X, Y) for nm, X, Y in synth_genome(n, cfg=1)]
eps, minPts, hic = pipe.MODES[1]
dataI, cut, cuts, steps = pipe.runSweepFast(fs, eps, minPts, cut=0)
records = {key: {"f": v["f"], "records": pipe._records(key, v["boxes"])} for key, v in dataI.items()}
print("candidates", sum(len(v["records"]) for v in records.values()), flush=True)
with tempfile.TemporaryDirectory() as td:
    pr = cProfile.Profile(); pr.enable()
    t = time.time()
    cModel.runStat(records, minPts, 0, 1, os.path.join(td, "o"), hic)
    pr.disable()
    print("runStat %.2f s" % (time.time() - t), flush=True)
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
