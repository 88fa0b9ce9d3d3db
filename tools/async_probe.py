import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import api
from cloops_amd.synth import synth_chrom
X, Y = synth_chrom(5000000, 248956422, 2000)
ch = api.Chromosome(X, Y)
for _ in range(3): ch.cluster("v2", 2000, 5, pinned=True)
T=[]
t=time.perf_counter(); ch.cluster_async("v2",2000,5); T.append(("async0",time.perf_counter()-t))
for k in range(6):
    t=time.perf_counter(); ch.cluster_async("v2",2000,5); T.append(("async",time.perf_counter()-t))
    t=time.perf_counter(); r=ch.wait(); T.append(("wait",time.perf_counter()-t))
t=time.perf_counter(); r=ch.wait(); T.append(("waitlast",time.perf_counter()-t))
for n,d in T: print(n, round(d*1e6), "us")
