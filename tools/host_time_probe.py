"""developer tool: where the HOST spends a sweep -- time inside the enqueue calls (cl_cluster_step_async), inside the waits, and
the rest (numpy aggregation, cut estimate).   python tools/host_time_probe.py [n_total]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import api, pipe
from cloops_amd.synth import synth_chrom, chrom_sizes

n_total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200000000
fs = []
for ci, (name, length, n) in enumerate(chrom_sizes(n_total)):
    X, Y = synth_chrom(n, length, 3000 + ci)
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
acc = {"enq": 0.0, "wait": 0.0, "n": 0}
orig_step, orig_wait = api.Chromosome.step_async, api.Chromosome.wait


def step(self, *a, **k):
    t0 = time.perf_counter()
    r = orig_step(self, *a, **k)
    acc["enq"] += time.perf_counter() - t0; acc["n"] += 1
    return r


def wait(self, *a, **k):
    t0 = time.perf_counter()
    r = orig_wait(self, *a, **k)
    acc["wait"] += time.perf_counter() - t0
    return r


api.Chromosome.step_async, api.Chromosome.wait = step, wait
eps, mps = [5000, 7500, 10000], [50, 40, 30, 20]
pipe.runSweepFast(fs, eps, mps, cut=0)
for rep in range(3):
    acc.update(enq=0.0, wait=0.0, n=0)
    t0 = time.perf_counter()
    res = pipe.runSweepFast(fs, eps, mps, cut=0)
    dt = time.perf_counter() - t0
    print("sweep %.1f ms: %d enqueues %.1f ms (%.0f us each), waits %.1f ms (summed over the pool's threads), steps %s" % (
        dt * 1e3, acc["n"], acc["enq"] * 1e3, acc["enq"] / max(1, acc["n"]) * 1e6, acc["wait"] * 1e3, ["%.1f" % (1e3 * s["wall_s"]) for s in res[3]]))
