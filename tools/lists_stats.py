"""Developer tool (devel library): event counts of the list kernels over the roofline replay (chr1, the sweep's 12 runs).
CLOOPS_DEVEL_LIB=1 python tools/lists_stats.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CLOOPS_DEVEL_LIB"] = "1"
os.environ["CLOOPS_DBG2"] = str(1 << 30)                  # the event counters are opt-in (same-address atomics)
import bench
from cloops_amd import api, _lib
from cloops_amd.synth import synth_chrom, chrom_sizes
CUTS_IN = [0, 4536, 6098, 6306, 5711, 3871, 5004, 5256, 5517, 4896, 5977, 6250]
name, length, n = chrom_sizes(200000000)[0]
X, Y = synth_chrom(n, length, 3000)
ch = api.Chromosome(X, Y)
ch.set_device_labels(False)
ch.set_count_thresholds([20, 30, 40, 50])
lib = _lib.load()
out = (ctypes.c_uint64 * 32)()
lib.cl_debug_lstats(out)
settings = [(eps, m, CUTS_IN[4 * i + j]) for i, eps in enumerate((5000, 7500, 10000)) for j, m in enumerate((50, 40, 30, 20))]
for eps, m, cut in settings:
    ch.cluster("v2", eps, m, cut)
    lib.cl_debug_lstats(out)
    v = list(out)
    print("(%5d, %2d, %4d): M %9d cores %9d walkers %8d | border: walkers %8d iters %9d (%.2f each, global %d) roots %.2f hinted %.3f owned %.3f | per wave: max iters %.1f" % (
        eps, m, cut, v[10], v[8], v[9], v[0], v[1], v[1] / max(1, v[0]), v[2], v[3] / max(1, v[0]), v[4] / max(1, v[0]), v[5] / max(1, v[0]), v[6] / max(1, v[7])))
    print("      union: cores %d walk rounds %.2f touches %.3f unites %.4f global-path %.4f jumps %.3f (per core)" % (
        v[12], v[13] / max(1, v[12]), v[14] / max(1, v[12]), v[15] / max(1, v[12]), v[16] / max(1, v[12]), v[17] / max(1, v[12])))
    print("      union: per wave max rounds %.1f; cores with > 2 / 4 / 8 rounds: %.3f %.3f %.3f" % (v[18] / max(1, v[19]), v[20] / max(1, v[12]), v[21] / max(1, v[12]), v[22] / max(1, v[12])))
