"""Per-step wall time of the mode-3 sweep on ONE chromosome (chr1 of the 200 M genome), cuts of the genome-wide chain forced:
what a rank of an 8-GPU run is bound by.  usage: python tools/one_chrom_steps.py [chromosome index ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import pipe
from cloops_amd.synth import synth_chrom, chrom_sizes

CUTS = [4536, 6098, 6306, 5711, 3871, 5004, 5256, 5517, 4896, 5977, 6250, 6428]
idx = [int(a) for a in sys.argv[1:]] or [0]
sizes = chrom_sizes(200000000)
fs = []
for ci in idx:
    name, length, n = sizes[ci]
    X, Y = synth_chrom(n, length, 3000 + ci)
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
for k in range(3):
    marks = []
    t0 = time.perf_counter()
    pipe.runSweepFast(fs, [5000, 7500, 10000], [50, 40, 30, 20], cut=0, forced_cuts=CUTS, log=lambda m: marks.append(time.perf_counter()))
    t1 = time.perf_counter()
    steps = [marks[0] - t0] + [b - a for a, b in zip(marks, marks[1:])]
    print("sweep %d over %s: %.1f ms; per step ms: %s; tail %.2f ms" % (k, [sizes[c][0] for c in idx], (t1 - t0) * 1e3, " ".join("%.2f" % (x * 1e3) for x in steps), (t1 - marks[-1]) * 1e3))
