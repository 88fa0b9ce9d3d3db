"""developer tool: the 200 M-PET mode-3 sweep with the region-query reuse (count cache) on / off, alternating inside ONE
process on one box (sweeps inside a process reproduce to +-0.5 %).   python tools/ab_reuse.py [reps] [n_total] [mode 3 | 4 | 5]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import pipe
from cloops_amd.synth import synth_chrom, chrom_sizes

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_total = int(float(sys.argv[2])) if len(sys.argv) > 2 else 200000000
fs = []
for ci, (name, length, n) in enumerate(chrom_sizes(n_total)):
    X, Y = synth_chrom(n, length, 3000 + ci)
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
MODES = {3: ([5000, 7500, 10000], [50, 40, 30, 20]), 4: ([2500, 5000, 7500, 10000], [30, 20]), 5: (list(range(1000, 10001, 1000)), [50, 30, 20, 10, 5])}
eps, mps = MODES[int(sys.argv[3]) if len(sys.argv) > 3 else 3]


def sweep(on):
    for f in fs:
        pipe.CACHE.get(f).chrom.set_count_reuse(on)
    t0 = time.perf_counter()
    r = pipe.runSweepFast(fs, eps, mps, cut=0)
    return time.perf_counter() - t0, r[1], [s.get("cut_out") for s in r[3]]


sweep(True); sweep(False)
tot = {True: [], False: []}
for k in range(reps):
    for on in (True, False):
        dt, cut, cuts = sweep(on)
        tot[on].append(dt)
for on in (True, False):
    v = sorted(tot[on])
    print("count reuse %-3s: median %.1f ms, min %.1f ms  %s" % ("on" if on else "off", 1e3 * v[len(v) // 2], 1e3 * v[0], ["%.1f" % (1e3 * x) for x in tot[on]]))
print("final cut", cut, cuts[:12])
