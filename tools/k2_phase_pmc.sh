#!/bin/bash
# developer tool: VALU / SALU / LDS wave-instructions of the region-query launches of the roofline replay, stopped after each phase
# (devel library, CLOOPS_DBG bits as in k2_phase_replay.sh).  Per launch: instructions per 64 PETs of the 16.4 M rows.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export CLOOPS_DEVEL_LIB=1 CLOOPS_REPLAY_ONLY=reuse
for DBG in ${@:-32 64 128 2048 0}; do
  rm -rf /tmp/k2pp
  CLOOPS_DBG=$DBG timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/k2pp -o p -- python $REPO/tools/k2_replay.py 1 > /tmp/k2pp.log 2>&1
  python3 - "$DBG" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/k2pp/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_region_" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
per = 16427116 / 64.0
print("dbg=%-5s " % sys.argv[1] + "  ".join("%s %s" % (k.replace("SQ_", ""), " ".join("%.0f" % (x / per) for x in v[:3])) for k, v in sorted(acc.items())))
PY
done
