"""per-sweep wall clock over many repetitions in ONE process (is the 217 / 231 ms bimodality per process or per sweep?)"""
import os, sys, time
if os.environ.get("TORCH_FIRST"):
    import torch
    print("torch", torch.__version__, "hip", torch.version.hip)
    if os.environ.get("TORCH_FIRST") == "2":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1)
        t = torch.ones(4, device="cuda:0"); dist.all_reduce(t); torch.cuda.synchronize(); print("rccl ok", t.tolist())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloops_amd import pipe
from cloops_amd.synth import synth_genome
fs = []
for name, X, Y in synth_genome(200000000, cfg=3):
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
if os.environ.get('SERIAL_ENQUEUE'):
    pipe.PARALLEL_ENQUEUE = False
ts = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    t0 = time.perf_counter()
    pipe.runSweepFast(fs, [5000, 7500, 10000], [50, 40, 30, 20], cut=0)
    ts.append(time.perf_counter() - t0)
print(" ".join("%.3f" % t for t in ts))
