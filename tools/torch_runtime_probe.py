import os, sys, time
if os.environ.get("WITH_TORCH") == "1":
    import torch
    torch.cuda.set_device(0)
    print("torch", torch.__version__, torch.version.hip)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from cloops_amd import api
from cloops_amd.synth import synth_chrom
X, Y = synth_chrom(5000000, 248956422, 2000)
ch = api.Chromosome(X, Y)
ch.set_profiling(True)
for it in range(4):
    res = ch.cluster("v2", 2000, 5, pinned=True)
tm = res.timing
print({k[3:]: round(v, 3) for k, v in tm.items() if k.startswith("ms_")})
# pipelined throughput
ch.set_profiling(False)
def run(n):
    ch.cluster_async("v2", 2000, 5, 0)
    for k in range(n):
        if k + 1 < n: ch.cluster_async("v2", 2000, 5, 0)
        ch.wait()
run(5)
t = time.perf_counter(); run(40); dt = time.perf_counter() - t
print("pipelined %.3f ms/step" % (dt / 40 * 1e3))
import subprocess
print(open("/proc/self/maps").read().count("libamdhip64"), [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
if os.environ.get("WITH_TORCH") == "1" and os.environ.get("WITH_PG") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    run(5)
    t = time.perf_counter(); run(40); dt = time.perf_counter() - t
    print("after init_process_group: pipelined %.3f ms/step" % (dt / 40 * 1e3))
    dist.barrier(); torch.cuda.synchronize()
    run(5)
    t = time.perf_counter(); run(40); dt = time.perf_counter() - t
    print("after first collective:   pipelined %.3f ms/step" % (dt / 40 * 1e3))
    x = torch.zeros(1000, device="cuda")
    run(5)
    t = time.perf_counter(); run(40); dt = time.perf_counter() - t
    print("after a torch allocation: pipelined %.3f ms/step" % (dt / 40 * 1e3))
    dist.destroy_process_group()
