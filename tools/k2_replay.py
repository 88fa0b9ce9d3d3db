"""The roofline leg of bench.py on its own: chr1 of the 200 M-PET genome (BASELINE.json configs[3]) alone on the GPU, the
12 (eps, minPts, cut) settings of the mode-3 chained sweep (cuts = the chain this genome produces), `reps` launches
each.  Run under rocprofv3 by tools/profile_bench.sh for the K2 kernel stats and the FETCH_SIZE / WRITE_SIZE passes.

    python tools/k2_replay.py [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import api
from cloops_amd.synth import synth_chrom, chrom_sizes

CUTS_IN = [0, 4536, 6098, 6306, 5711, 3871, 5004, 5256, 5517, 4896, 5977, 6250]      # cut_in of the 12 runs (bench.py config.cuts, shifted)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
name, length, n = chrom_sizes(200000000)[0]
X, Y = synth_chrom(n, length, 3000)
ch = api.Chromosome(X, Y)
ch.set_profiling(True)
ch.set_device_labels(False)
tot_b = tot_ms = 0.0
k = 0
for eps in (5000, 7500, 10000):
    for m in (50, 40, 30, 20):
        for r in range(reps):
            ch.cluster_async("v2", eps, m, CUTS_IN[k], want_labels=False, want_boxes=False)
            tm = ch.wait().timing
            tot_b += tm["n_in"] * 12 + tm["n_strips"] * 4
            tot_ms += max(tm["ms_region"] - tm["ms_bracket"], 1e-6)
        k += 1
print("K2 over %d launches: %.1f GB/s algorithmic = %.2f %% of 8 TB/s, avg %.1f us" % (12 * reps, tot_b / tot_ms / 1e6, tot_b / tot_ms / 1e6 / 80, tot_ms / (12 * reps) * 1e3))
