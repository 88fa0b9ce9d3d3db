#!/bin/bash
# instruction counts of K2 per ablation level (developer tool): usage tools/k2_pmc.sh
cd /tmp && export TMPDIR=/tmp
for d in 32 64 128 0; do
  rm -rf /tmp/pmc_$d
  CLOOPS_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_$d -o p -- python $GRAFT_REPO_ROOT/tools/quick_timing.py 5e6 2000 5 v2 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_$d/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_region_count" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("dbg=$d", {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
done
