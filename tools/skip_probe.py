"""developer tool (developer library, CLOOPS_DEVEL_LIB=1): the sweep with kernels left out (CLOOPS_SKIP=<mask>; results invalid, the
chain's cuts forced) -- what a kernel costs the SWEEP.   CLOOPS_SKIP=1 python tools/skip_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import pipe
from cloops_amd.synth import synth_chrom, chrom_sizes

fs = []
for ci, (name, length, n) in enumerate(chrom_sizes(200000000)):
    X, Y = synth_chrom(n, length, 3000 + ci)
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
eps, mps = [5000, 7500, 10000], [50, 40, 30, 20]
forced = [4536, 6098, 6306, 5711, 3871, 5004, 5256, 5517, 4896, 5977, 6250, 6428]
pipe.runSweepFast(fs, eps, mps, cut=0, forced_cuts=forced)
ts = []
for rep in range(4):
    t0 = time.perf_counter()
    pipe.runSweepFast(fs, eps, mps, cut=0, forced_cuts=forced)
    ts.append(time.perf_counter() - t0)
print("CLOOPS_SKIP=%-4s sweep %.1f ms (min of %s)" % (os.environ.get("CLOOPS_SKIP", "0"), 1e3 * min(ts), ["%.1f" % (1e3 * t) for t in ts]))
