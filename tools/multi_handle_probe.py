"""Throughput of independent runs on ONE GPU when H handles (own workspace + streams) of the same chromosome
take turns (developer probe).  usage: multi_handle_probe.py n H inflight want_labels"""
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(25, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import api
from cloops_amd.synth import synth_chrom
n = int(float(sys.argv[1])); H = int(sys.argv[2]); F = int(sys.argv[3]); WL = bool(int(sys.argv[4]))
X, Y = synth_chrom(n, 248956422, 2000)
hs = [api.Chromosome(X, Y) for _ in range(H)]
def run(steps):
    q = []                                  # FIFO of handles with a run in flight
    issued = done = 0
    res = None
    while done < steps:
        while issued < steps and len(q) < H * F:
            h = hs[issued % H]; h.cluster_async("v2", 2000, 5, 0, want_labels=WL); q.append(h); issued += 1
        res = q.pop(0).wait(); done += 1
    return res
run(6)
print("warm", flush=True)
t0 = time.perf_counter(); steps = 40; res = run(steps); dt = time.perf_counter() - t0
print("H=%d F=%d labels=%d: %.3f ms/step  %.2f G PETs/s  (K=%d)" % (H, F, WL, dt / steps * 1e3, n * steps / dt / 1e9, res.n_clusters), flush=True)
