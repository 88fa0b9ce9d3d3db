import sys, time, os
if "--torch" in sys.argv:
    import torch
    torch.cuda.set_device(0); x = torch.zeros(4, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import api
from cloops_amd.synth import synth_chrom
X, Y = synth_chrom(5000000, 248956422, 2000)
ch = api.Chromosome(X, Y); ch.set_profiling(True)
def run(n):
    ch.cluster_async("v2", 2000, 5)
    for k in range(n):
        if k + 1 < n: ch.cluster_async("v2", 2000, 5)
        r = ch.wait()
    return r
run(3)
for n in (10, 10, 50):
    t = time.perf_counter(); r = run(n); dt = time.perf_counter() - t
    print("steps", n, "ms/step", round(dt / n * 1e3, 3), {k: round(v, 3) for k, v in r.timing.items() if k.startswith("ms_")})
