"""Where does the first sweep of a process go?  usage: python tools/first_sweep_probe.py [warm]
warm = 1: one tiny clustering run + sweep on a 20 000-PET chromosome first (kernel code objects loaded, rocPRIM initialised)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import pipe
from cloops_amd.synth import synth_chrom, chrom_sizes

warm = len(sys.argv) > 1 and sys.argv[1] == "1"
sizes = chrom_sizes(200000000)
t0 = time.perf_counter()
fs = []
for ci, (name, length, n) in enumerate(sizes):
    X, Y = synth_chrom(n, length, 3000 + ci)
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
print("synthesis + upload %.2f s" % (time.perf_counter() - t0))
if warm:
    X, Y = synth_chrom(20000, 1000000, 1)
    f = pipe.CACHE.put_arrays("w-w", X, Y)
    t0 = time.perf_counter()
    pipe.runSweepFast([f], [5000, 7500], [50, 40], cut=0)
    print("tiny warm-up sweep %.3f s" % (time.perf_counter() - t0))
if len(sys.argv) > 1 and sys.argv[1] == "2":
    # keep the GPU busy for ~0.6 s on ONE other handle (clock ramp?), touching none of the sweep's handles
    from cloops_amd import api
    X, Y = synth_chrom(3000000, 46709983, 77)
    ch = api.Chromosome(X, Y)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:
        ch.cluster("v2", 5000, 30, 0, want_labels=False, want_boxes=False)
    ch.close()
from cloops_amd import api as _api
_acc = {"enq": 0.0, "wait": 0.0, "res": 0.0}
def _wrap(name, key):
    f = getattr(_api.Chromosome, name)
    def g(self, *a, **kw):
        t = time.perf_counter()
        try:
            return f(self, *a, **kw)
        finally:
            _acc[key] += time.perf_counter() - t
    setattr(_api.Chromosome, name, g)
_wrap("step_async", "enq"); _wrap("wait", "wait"); _wrap("step_result", "res")
# host gap between the steps: from the end of a step's last step_result to the first step_async of the next step
_ev = []
def _stamp(name, when):
    f = getattr(_api.Chromosome, name)
    def g(self, *a, **kw):
        if when == "before":
            _ev.append((time.perf_counter(), name))
        try:
            return f(self, *a, **kw)
        finally:
            if when == "after":
                _ev.append((time.perf_counter(), name))
    setattr(_api.Chromosome, name, g)
_stamp("step_async", "before"); _stamp("step_result", "after")
for k in range(int(os.environ.get("N_SWEEPS", "3"))):
    t0 = time.perf_counter()
    marks = []
    pipe.runSweepFast(fs, [5000, 7500, 10000], [50, 40, 30, 20], cut=0, log=lambda m: marks.append(time.perf_counter()))
    t1 = time.perf_counter()
    steps = [marks[0] - t0] + [b - a for a, b in zip(marks, marks[1:])]
    evs = sorted(_ev); gaps = [b[0] - a[0] for a, b in zip(evs, evs[1:]) if a[1] == "step_result" and b[1] == "step_async"]
    print("   host gaps between steps (last step_result -> first step_async), ms: %s; sum %.1f ms" % (" ".join("%.2f" % (x * 1e3) for x in gaps), sum(gaps) * 1e3))
    del _ev[:]
    print("   host: enqueue %.1f ms, wait %.1f ms (summed over the pool's threads), step_result %.1f ms" % (_acc["enq"] * 1e3, _acc["wait"] * 1e3, _acc["res"] * 1e3))
    for kk in _acc:
        _acc[kk] = 0.0
    print("sweep %d: %.3f s; per step ms: %s; tail (candidates) %.1f ms" % (k, t1 - t0, " ".join("%.1f" % (x * 1e3) for x in steps), (t1 - marks[-1]) * 1e3))
