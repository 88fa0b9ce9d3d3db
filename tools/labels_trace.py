"""Developer tool: the label-inclusive form of a run on chr1 of the 200 M genome, for rocprofv3 --kernel-trace --stats:
    python tools/labels_trace.py [mask|pairs|rows]     (four runs of the mode-3 settings, labels to the host every run)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import api
from cloops_amd.synth import synth_chrom, chrom_sizes
mode = sys.argv[1] if len(sys.argv) > 1 else "mask"
name, length, n = chrom_sizes(200000000)[0]
X, Y = synth_chrom(n, length, 3000)
ch = api.Chromosome(X, Y)
ch.sweep_plan([7500], [50, 40, 30, 20])
runs = [(7500, 50, 0), (7500, 40, 4536), (7500, 30, 5004), (7500, 20, 5256)]
for rep in range(3):
    for eps, m, cut in runs:
        if mode == "mask":
            ch.cluster_rowmask_async("v2", eps, m, cut); k = len(ch.wait_rowmask()[2])
        elif mode == "pairs":
            ch.cluster_pairs_async("v2", eps, m, cut); k = len(ch.wait_pairs()[1])
        else:
            ch.cluster_async("v2", eps, m, cut, want_labels=True); k = int((ch.wait().labels >= 0).sum())
print(mode, "labelled PETs of the last run:", k)
