"""Where the host time of runSweepFast goes (per-function wall time summed over the pool threads)."""
import os, sys, time, threading, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import pipe, api
from cloops_amd.synth import synth_genome

n_total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 5
MODES = {3: ([5000, 7500, 10000], [50, 40, 30, 20]), 5: (list(range(1000, 10001, 1000)), [50, 30, 20, 10, 5])}
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)
lock = threading.Lock()

def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t
            with lock:
                acc[name] += dt; cnt[name] += 1
    setattr(obj, name, w)

for nm in ("wait", "cluster_async", "dist_stats", "dist_sqdev", "dist_hist"):
    wrap(api.Chromosome, nm)
for nm in ("_boxes_classified", "_combine_steps", "_select_kth"):
    wrap(pipe, nm)
fs = []
for name, X, Y in synth_genome(n_total, cfg=mode):
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
eps, minPts = MODES[mode]
pipe.runSweepFast(fs, eps[:1], minPts[:1], cut=0)
acc.clear(); cnt.clear()
t0 = time.perf_counter()
dataI, cut, cuts, steps = pipe.runSweepFast(fs, eps, minPts, cut=0)
dt = time.perf_counter() - t0
print("sweep %.3f s, %d steps, %d candidates" % (dt, len(steps), sum(len(v["boxes"]) for v in dataI.values())))
for k in sorted(acc, key=lambda k: -acc[k]):
    print("  %-20s %8.3f s  (%d calls)" % (k, acc[k], cnt[k]))
print("clusters per step:", [s["n_inter"] + s["n_self"] for s in steps][:12])
