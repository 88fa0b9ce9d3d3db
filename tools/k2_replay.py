"""The roofline leg of bench.py on its own: chr1 of the 200 M-PET genome (BASELINE.json configs[3]) alone on the GPU, the
12 (eps, minPts, cut) runs of the mode-3 chained sweep in the sweep's order (cuts = the chain this genome produces), `passes`
times with the region query re-used inside an eps, then as often without.  Run under rocprofv3 by tools/profile_bench.sh for
the kernel stats and the FETCH_SIZE / WRITE_SIZE passes.

    python tools/k2_replay.py [passes]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cloops_amd import api
from cloops_amd.synth import synth_chrom, chrom_sizes

CUTS_IN = [0, 4536, 6098, 6306, 5711, 3871, 5004, 5256, 5517, 4896, 5977, 6250]      # cut_in of the 12 runs (bench.py config.cuts, shifted)
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
name, length, n = chrom_sizes(200000000)[0]
X, Y = synth_chrom(n, length, 3000)
ch = api.Chromosome(X, Y)
ch.set_device_labels(False)
settings = [(eps, m, CUTS_IN[4 * i + j]) for i, eps in enumerate((5000, 7500, 10000)) for j, m in enumerate((50, 40, 30, 20))]
rep = bench.k2_replay(ch, settings, [20, 30, 40, 50], passes)
blk = bench.roofline_block(rep, n)
for a, f in list(zip(rep["reuse"], rep["full"]))[:12]:
    print("run (%5d, %2d, %4d) mode %d: sort bracket %6.1f us (full: %6.1f), region %6.1f us (full: %6.1f)" % (
        a[0], a[1], a[2], a[4], 1e3 * a[3]["ms_sort"], 1e3 * f[3]["ms_sort"], 1e3 * a[3]["ms_region"], 1e3 * f[3]["ms_region"]), file=sys.stderr)
print(json.dumps(blk, indent=1))
print("K2 amortised over %d runs: %.1f GB/s algorithmic = %.2f %% of 8 TB/s (first run of an eps %.1f us, band %.1f us + carry %.1f us); "
      "every run its own query: %.2f %%, %.1f us" % (blk["launches"], blk["achieved"], 100 * blk["frac"], 1e3 * blk["first_run_avg_launch_ms"],
                                                      1e3 * blk["band_avg_launch_ms"], 1e3 * blk["carry_avg_ms"], 100 * blk["full_query"]["frac"],
                                                      1e3 * blk["full_query"]["avg_launch_ms"]))
