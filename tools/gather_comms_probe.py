"""developer tool: cl_comm_gather_device at world 1 from one process, communicator after communicator (the rate of the
device-to-host copies as the 1st, 2nd, 3rd ... communicator of a process sees it).  python tools/gather_comms_probe.py [keep]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from cloops_amd.comm import Comm
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
rows = [158000] * 23
ptrs = []
for k in rows:
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), k * 16) == 0
    hip.hipMemset(p, 1, k * 16)
    ptrs.append(p.value)
hip.hipDeviceSynchronize()
held = []
for i in range(5):
    c = Comm(0, 1, 0)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        n = sum(len(t) for t in c.gather_device(ptrs, rows, dst=0, copy=False))
        ts.append(time.perf_counter() - t0)
    print("communicator %d: rows %d, gathers %s ms" % (i, n, " ".join("%.2f" % (t * 1e3) for t in ts)), flush=True)
    if keep:
        held.append(c)
    else:
        c.close()
