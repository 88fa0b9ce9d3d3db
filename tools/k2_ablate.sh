#!/bin/bash
# developer tool: VALU / SALU / LDS instruction counts of K2 per phase (ablations of the devel library), old and new kernel
# usage (on the GPU box): tools/k2_ablate.sh [eps minPts cut]
cd /tmp && export TMPDIR=/tmp
export CLOOPS_DEVEL_LIB=1 DENSE_NOLABELS=1
ARGS="${1:-7500} ${2:-30} ${3:-5004} 1"
for OLD in 1; do
  KN=k_region_core
  for DBG in ${ABL_LIST:-32 64 128 0}; do
    rm -rf /tmp/k2abl
    CLOOPS_DBG=$DBG timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/k2abl -o p -- python /root/repo/tools/dense_run.py $ARGS > /dev/null 2>&1
    python3 - "$KN" "$OLD" "$DBG" <<'PY'
import csv, glob, collections, sys
kn, old, dbg = sys.argv[1:4]
f = glob.glob("/tmp/k2abl/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if kn in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = max(m.get("SQ_WAVES", 1), 1)
print("old=%s stop_after=%-3s  VALU/wave %7.1f  SALU/wave %6.1f  LDS/wave %6.1f  busy_us %6.1f  wave_kcyc %5.1f" % (
    old, {"32": "stg", "64": "ph0", "128": "ph1", "0": "all"}.get(dbg, dbg), m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w,
    m.get("SQ_BUSY_CYCLES", 0) / 32 / 2400.0, m.get("SQ_WAVE_CYCLES", 0) * 4 / w / 1e3))
PY
  done
done
