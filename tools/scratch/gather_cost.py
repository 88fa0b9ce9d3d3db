"""cost of the sweep's final exchange (dist.gather_tables over RCCL) at world size 1 -- the per-rank staging costs are the same at any N"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from cloops_amd import pipe
from cloops_amd.dist import gather_tables, make_allsum
from cloops_amd.synth import synth_genome
fs = []
for name, X, Y in synth_genome(200000000, cfg=3):
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
g = dist.new_group(backend="gloo")
allsum = make_allsum(device=None, group=g)
dev = torch.device("cuda:0")
for rep in range(4):
    t0 = time.perf_counter()
    dataI, cut, cuts, steps = pipe.runSweepFast(fs, [5000, 7500, 10000], [50, 40, 30, 20], cut=0, allsum=allsum)
    t1 = time.perf_counter()
    rows = [v["boxes"] for v in dataI.values() if len(v["boxes"])]
    tab = rows
    t2 = time.perf_counter()
    out = gather_tables(tab, device=dev, dst=0)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("sweep %.3f s | concat %.1f ms | gather_tables %.1f ms (%d rows)" % (t1 - t0, (t2 - t1) * 1e3, (t3 - t2) * 1e3, sum(len(t) for t in out)))
dist.destroy_process_group()
