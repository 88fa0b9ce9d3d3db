"""Developer tool: the device-resident gather at world size 1 (cl_comm_gather_device) against one plain device-to-host copy of the
same bytes.  python tools/gather_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import pipe, comm
from cloops_amd.synth import synth_chrom, chrom_sizes
pipe.CACHE.clear()
fs = []
for ci, (name, length, n) in enumerate(chrom_sizes(int(sys.argv[1]) if len(sys.argv) > 1 else 100000000)):
    X, Y = synth_chrom(n, length, 3000 + ci)
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
c = comm.Comm(0, 1, 0)
d = pipe.runSweepFast(fs, [5000, 7500], [50, 30], cut=0, finish_device=True)[0]
ptrs, rows = [v["dev_rows"] for v in d.values()], [v["n_rows"] for v in d.values()]
tot = sum(rows)
for k in range(4):
    t0 = time.perf_counter()
    out = c.gather_device(ptrs, rows, dst=0, copy=False)
    print("gather_device: %d rows (%.1f MB) in %.2f ms" % (len(out[0]), tot * 16 / 1e6, 1e3 * (time.perf_counter() - t0)))
hip = ctypes.CDLL("libamdhip64.so")
dev, host = ctypes.c_void_p(), ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(tot * 16)); hip.hipHostMalloc(ctypes.byref(host), ctypes.c_size_t(tot * 16), 0)
for k in range(3):
    t0 = time.perf_counter()
    hip.hipMemcpy(host, dev, ctypes.c_size_t(tot * 16), 2)
    print("one hipMemcpy D2H of the same bytes: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
c.close()
