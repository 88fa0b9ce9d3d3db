"""BUILD CONTAINER ONLY (needs the reference checkout at /root/reference): the speed of the reference's own clustering classes
in CPython, one core -- the number bench.py's cpu_baseline carries as `reference_python_pets_per_s_per_core` (the C oracle that
bench.py times on the GPU box is a port, ~20x faster than this).  Writes profiles/reference_python_speed.json.

    python tools/measure_reference_python.py
"""
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refload
import golden_util as G
from cloops_amd.synth import synth_chrom

cls = refload.ref_classes()
rows = []


def run(name, X, Y, variant, eps, minPts):
    mat = np.stack([np.arange(len(X)), X, Y], 1).astype(np.int64)
    t0 = time.perf_counter()
    db = cls[variant](mat, eps, minPts)
    dt = time.perf_counter() - t0
    rows.append({"workload": name, "variant": variant, "eps": eps, "minPts": minPts, "pets": int(len(X)), "seconds": round(dt, 3),
                 "pets_per_s": round(len(X) / dt, 1), "labelled": len(db.labels)})
    print(rows[-1], flush=True)


X, Y = G.chr21_xy()
for eps in (500, 1000, 2000):
    for v in ("v2", "v1", "block"):
        run("examples/GSM1872886 chr21 (99 674 PETs)", X, Y, v, eps, 5)
X, Y = synth_chrom(400000, 248956422 // 20, 3000)          # the density of chr1 of the 200 M genome on 1/20 of its length
for eps, m in ((5000, 50), (7500, 30)):
    run("synthetic 400 k PETs at the density of chr1 of the 200 M-PET genome", X, Y, "v2", eps, m)
v2 = [r for r in rows if r["variant"] == "v2"]
dense = [r for r in v2 if r["workload"].startswith("synthetic")]
out = {"what": "cLoops/cDBSCAN2.py (production class, .iteritems -> .items in memory), cLoops/cDBSCAN.py, cLoops/blockDBSCAN.py imported from the "
               "reference checkout and timed in CPython %s on ONE core of the build container (%s, %d cores visible)" % (
                   platform.python_version(), platform.processor() or platform.machine(), os.cpu_count() or 1),
       "runs": rows,
       "reference_python_pets_per_s_per_core": round(sum(r["pets"] for r in dense) / sum(r["seconds"] for r in dense), 1),
       "reference_python_pets_per_s_per_core_note": "cDBSCAN2 on the synthetic workload at the headline bench's own density and settings (mode-3 eps / minPts)",
       "reference_python_pets_per_s_per_core_chr21_example": round(sum(r["pets"] for r in v2 if r not in dense) / sum(r["seconds"] for r in v2 if r not in dense), 1)}
json.dump(out, open(os.path.join(ROOT, "profiles", "reference_python_speed.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "runs"}, indent=1))
