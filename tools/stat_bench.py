import os, sys, tempfile, time, faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("DBG_T", "50")), exit=True)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cloops_amd import pipe, cModel
from cloops_amd.synth import synth_genome
t=time.time()
fs = [pipe.CACHE.put_arrays("%s-%s" % (n, n), X, Y) for n, X, Y in synth_genome(int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000, cfg=1)]
eps, minPts, hic = pipe.MODES[1]
dataI, cut, cuts, steps = pipe.runSweepFast(fs, eps, minPts, cut=0)
print("sweep done", time.time()-t, sum(len(v["boxes"]) for v in dataI.values()), flush=True)
records = {key: {"f": v["f"], "records": pipe._records(key, v["boxes"])} for key, v in dataI.items()}
print("records done", time.time()-t, flush=True)
with tempfile.TemporaryDirectory() as td:
    cModel.runStat(records, minPts, 0, 1, os.path.join(td, "o"), hic)
print("stat done", time.time()-t, flush=True)
