import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
from cloops_amd.dist import gather_tables
tab = np.random.randint(0, 1<<28, (78806,5)).astype(np.int32)
dev = torch.device("cuda",0)
for it in range(5):
    torch.cuda.synchronize(); t=time.perf_counter(); out = gather_tables(tab, device=dev); torch.cuda.synchronize(); print("gather", round((time.perf_counter()-t)*1e3,3),"ms")
for it in range(3):
    t=time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); print("barrier", round((time.perf_counter()-t)*1e3,3),"ms")
dist.destroy_process_group()
