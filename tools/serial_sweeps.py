"""developer tool: the 12-run sweep of every chromosome ALONE, one after the other (fixed cut chain of the genome-wide sweep),
to compare the sum of solo GPU times with the concurrent sweep's wall clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import api
from cloops_amd.synth import synth_chrom, chrom_sizes

CUTS_IN = [0, 4536, 6098, 6306, 5711, 3871, 5004, 5256, 5517, 4896, 5977, 6250]
sizes = chrom_sizes(200000000)
chroms = []
for ci, (name, length, n) in enumerate(sizes):
    X, Y = synth_chrom(n, length, 3000 + ci)
    ch = api.Chromosome(X, Y)
    ch.set_device_labels(False)
    chroms.append(ch)
for rep in range(2):
    t0 = time.perf_counter()
    per = []
    for ch in chroms:
        t1 = time.perf_counter()
        ch.cand_reset()
        k = 0
        for eps in (5000, 7500, 10000):
            for m in (50, 40, 30, 20):
                ch.step_async("v2", eps, m, CUTS_IN[k], k, 1024 if k else -1)
                ch.wait()
                ch.step_result()
                k += 1
        per.append(time.perf_counter() - t1)
    print("rep %d: serial sum over 23 chromosomes %.3f s (chr1 %.1f ms, chr21 %.1f ms)" % (rep, time.perf_counter() - t0, per[0] * 1e3, per[20] * 1e3))
