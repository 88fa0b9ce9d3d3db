"""Wall-clock of a whole (eps x minPts) sweep over a synthetic genome on ONE GPU, with the
chained cut of cLoops/pipe.py:247-275 (runSweep) -- BASELINE.json metric part 2.

    python tools/sweep_bench.py [n_total] [mode] [--oracle-n N]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import pipe
from cloops_amd.synth import synth_genome

MODES = {1: ([500, 1000, 2000], [5]), 2: ([1000, 2000, 5000], [5]), 3: ([5000, 7500, 10000], [50, 40, 30, 20]),
         4: ([2500, 5000, 7500, 10000], [30, 20]),          # cLoops/pipe.py:329-344
         5: (list(range(1000, 10001, 1000)), [50, 30, 20, 10, 5])}      # BASELINE.json configs[4]: dense user sweep
FAST_ONLY = "--fast-only" in sys.argv
if FAST_ONLY:
    sys.argv.remove("--fast-only")

if os.environ.get("SWEEP_THREADS"):
    pipe.SWEEP_THREADS = int(os.environ["SWEEP_THREADS"])
n_total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eps, minPts = MODES[mode]
t0 = time.perf_counter()
fs = []
npets = 0
for name, X, Y in synth_genome(n_total, cfg=mode):
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
    npets += len(X)
t_up = time.perf_counter() - t0
print("generated + uploaded %d PETs on %d chromosomes in %.1f s" % (npets, len(fs), t_up))
for rep in range(1 if FAST_ONLY else 0, 3):
    fast = rep > 0
    t0 = time.perf_counter()
    dataI, cut, cuts, steps = (pipe.runSweepFast if fast else pipe.runSweep)(fs, eps, minPts, cut=0)
    dt = time.perf_counter() - t0
    clustered = sum(s.get("n_in", 0) for s in steps)
    ncand = sum(len(v["boxes" if fast else "records"]) for v in dataI.values())
    print("rep %d (%s): mode %d sweep (%d steps, chained cut) %.3f s; final cut %d; %d candidate loops; cuts %s" % (
        rep, "runSweepFast" if fast else "runSweep", mode, len(steps), dt, cut, ncand, [s.get("cut_out") for s in steps]))
    print("   PETs entering DBSCAN summed over steps: %s -> %.2f G PETs/s" % (clustered, clustered / dt / 1e9 if clustered else 0))
