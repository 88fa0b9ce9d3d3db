"""Developer tool: host turnaround at the step boundaries of the 200 M mode-3 sweep -- from the moment the last chromosome's wait
returned (the GPU has nothing queued from then on) to the first enqueue of the next step.  python tools/step_gap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import pipe
from cloops_amd.synth import synth_chrom, chrom_sizes
fs = []
for ci, (name, length, n) in enumerate(chrom_sizes(200000000)):
    X, Y = synth_chrom(n, length, 3000 + ci)
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
for rep in range(3):
    pipe.STEP_TRACE = []
    t0 = time.perf_counter()
    pipe.runSweepFast(fs, [5000, 7500, 10000], [50, 40, 30, 20], cut=0)
    t1 = time.perf_counter()
    tr = pipe.STEP_TRACE
    gaps = []
    for k in range(11):
        last_wait = max(t for s, what, t in tr if s == k and what == "waited")
        first_wait = min(t for s, what, t in tr if s == k and what == "waited")
        nxt = [t for s, what, t in tr if s == k + 1 and what == "enqueue"][0]
        gaps.append(((nxt - last_wait) * 1e3, (last_wait - first_wait) * 1e3))
    print("sweep %d: %.1f ms; boundary gaps ms (last wait -> next first enqueue): %s; total %.2f ms" % (
        rep, (t1 - t0) * 1e3, " ".join("%.2f" % g for g, _ in gaps), sum(g for g, _ in gaps)))
    for what in ("collected", "reduced", "stats", "cut"):
        d = []
        for k in range(11):
            last_wait = max(t for s_, w, t in tr if s_ == k and w == "waited")
            tt = [t for s_, w, t in tr if s_ == k and w == what]
            d.append((tt[0] - last_wait) * 1e3 if tt else float("nan"))
        print("         last wait -> %-9s ms: %s" % (what, " ".join("%.2f" % x for x in d)))
    print("         spread of the waits inside a step (first .. last return) ms: %s" % " ".join("%.2f" % w for _, w in gaps))
