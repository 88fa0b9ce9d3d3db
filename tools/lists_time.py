"""Developer tool (devel library): k_classify / k_make_lists of one chr1 run timed on their own (cl_debug_time_lists), under the
CLOOPS_DBG2 ablation of the environment.  python tools/lists_time.py [eps minPts cut]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CLOOPS_DEVEL_LIB"] = "1"
from cloops_amd import api, _lib
from cloops_amd.synth import synth_chrom, chrom_sizes
eps, m, cut = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (7500, 30, 5004)
name, length, n = chrom_sizes(200000000)[0]
X, Y = synth_chrom(n, length, 3000)
ch = api.Chromosome(X, Y)
ch.set_device_labels(False)
saved = os.environ.pop("CLOOPS_DBG2", None)               # the run itself without ablation
lib = _lib.load()
lib.cl_debug_time_lists.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
ch.cluster("v2", eps, 50, 0)
ch.cluster("v2", eps, m, cut)
L4 = os.environ.get("CLOOPS_TRAVERSAL", "4") == "4"
KA, KB = (6, 7) if L4 else (0, 1)
for abl in [0] + [int(a) for a in (saved or "").split(",") if a]:
    os.environ["CLOOPS_DBG2"] = str(abl)
    out = []
    ms = ctypes.c_float(0)
    _lib.check(lib.cl_debug_time_lists(ch._h, KA, 20, ctypes.byref(ms)))
    out.append(ms.value * 1e3)
    os.environ["CLOOPS_DBG2"] = "0"                       # a valid set of masks for k_make_lists again
    _lib.check(lib.cl_debug_time_lists(ch._h, KA, 1, ctypes.byref(ms)))
    os.environ["CLOOPS_DBG2"] = str(abl)
    _lib.check(lib.cl_debug_time_lists(ch._h, KB, 20, ctypes.byref(ms)))
    out.append(ms.value * 1e3)
    if os.environ.get("CLOOPS_TIME_BORDER"):
        for which, name in ((4, "k_border_w"), (5, "k_final_lists"), (3, "chain+union+flatten")):
            _lib.check(lib.cl_debug_time_lists(ch._h, which, 10, ctypes.byref(ms)))
            out.append(ms.value * 1e3)
        print("dbg2 %7d: k_classify %.1f us  k_make_lists %.1f us  k_border_w %.1f us  k_final_lists %.1f us  chain+union+flatten %.1f us" % (abl, out[0], out[1], out[2], out[3], out[4]), flush=True)
        continue
    print("dbg2 %7d: k_classify %.1f us  k_make_lists %.1f us" % (abl, out[0], out[1]), flush=True)
