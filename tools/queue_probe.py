"""developer tool: every form of the path under the hardware-queue count given in the environment (GPU_MAX_HW_QUEUES): the 5 M
single-run form of bench.py (one handle, two result slots, labels + table to the host), and on a 23-chromosome genome the
sweep's own form (pipe.runSweepFast) and the label-inclusive form (labels + tables of every run on the host).
    python tools/queue_probe.py [label] [n_total] [streams]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cloops_amd import api, pipe
from cloops_amd.synth import synth_chrom, chrom_sizes

label = sys.argv[1] if len(sys.argv) > 1 else ""
n_total = int(float(sys.argv[2])) if len(sys.argv) > 2 else 200000000
if len(sys.argv) > 3:
    pipe.SWEEP_STREAMS = int(sys.argv[3])
X, Y = synth_chrom(5000000, 248956422, 2000)
ch = api.Chromosome(X, Y)
ch.set_layout_reuse(False)


def run(n):
    ch.cluster_async("v2", 2000, 5, 0)
    for k in range(n):
        if k + 1 < n:
            ch.cluster_async("v2", 2000, 5, 0)
        ch.wait()


run(3)
t0 = time.perf_counter(); run(20); t5 = (time.perf_counter() - t0) / 20
ch.close()
fs = []
for ci, (name, length, n) in enumerate(chrom_sizes(n_total)):
    Xc, Yc = synth_chrom(n, length, 3000 + ci)
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), Xc, Yc))
eps, mps = [5000, 7500, 10000], [50, 40, 30, 20]
pipe.runSweepFast(fs, eps, mps, cut=0)
t0 = time.perf_counter()
for _ in range(3):
    res = pipe.runSweepFast(fs, eps, mps, cut=0)
ts = (time.perf_counter() - t0) / 3
steps = res[3]
rs = sorted((pipe.CACHE.get(f) for f in fs), key=lambda r: -len(r))
for r in rs:
    r.chrom.set_device_labels(True)


def sweep_labels():
    for st in steps:
        for r in rs:
            r.chrom.cluster_async("v2", st["eps"], st["minPts"], st["cut_in"], want_labels=True, want_boxes=True)
        for r in rs:
            r.chrom.wait()


sweep_labels()
t0 = time.perf_counter(); sweep_labels(); sweep_labels(); tl = (time.perf_counter() - t0) / 2
print("%-28s 5M run %.3f ms | %s sweep %.1f ms | with labels %.1f ms" % (label, t5 * 1e3, "%dM" % (n_total // 1000000), ts * 1e3, tl * 1e3))
