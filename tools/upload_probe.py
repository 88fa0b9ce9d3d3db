"""Developer tool: the PCIe-inclusive figure of the headline workload: upload of the 23 chromosomes of the 200 M-PET
genome from host arrays (cl_chrom_create: X, Y cross PCIe once per chromosome, cLoops/io.py:206-217 re-reads its .jd every step) + the
dataset's one sweep.   python tools/upload_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import pipe
from cloops_amd.synth import synth_chrom, chrom_sizes

data = []
for ci, (name, length, n) in enumerate(chrom_sizes(200000000)):
    X, Y = synth_chrom(n, length, 3000 + ci)
    data.append((name, np.ascontiguousarray(X), np.ascontiguousarray(Y)))
eps, mps = [5000, 7500, 10000], [50, 40, 30, 20]
from cloops_amd import api
for rep in range(3):
    # the boundary alone: cl_chrom_create of every chromosome from (pageable) host arrays -- copy, distance histogram, extent statistics
    t0 = time.perf_counter()
    hs = [api.Chromosome(X, Y) for _, X, Y in data]
    t1 = time.perf_counter()
    for h in hs:
        h.close()
    print("rep %d: cl_chrom_create x 23: %.3f s = %.1f GB/s of coordinates" % (rep, t1 - t0, sum(X.nbytes + Y.nbytes for _, X, Y in data) / 1e9 / (t1 - t0)), flush=True)
for rep in range(3):
    pipe.CACHE.clear()
    t0 = time.perf_counter()
    fs = [pipe.CACHE.put_arrays("%s-%s" % (nm, nm), X, Y) for nm, X, Y in data]
    t1 = time.perf_counter()
    res = pipe.runSweepFast(fs, eps, mps, cut=0)
    t2 = time.perf_counter()
    pets = sum(s["n_in"] for s in res[3])
    print("rep %d: upload of %.2f GB %.3f s (%.1f GB/s), first sweep %.3f s, together %.3f s = %.2f G PETs/s entering DBSCAN (PCIe-inclusive)" % (
        rep, sum(X.nbytes + Y.nbytes for _, X, Y in data) / 1e9, t1 - t0, sum(X.nbytes + Y.nbytes for _, X, Y in data) / 1e9 / (t1 - t0), t2 - t1, t2 - t0, pets / (t2 - t0) / 1e9), flush=True)
