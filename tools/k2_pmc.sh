#!/bin/bash
# developer tool: SQ counters of the K2 kernels, per launch.  usage: tools/k2_pmc.sh "<cmd>" [kernel-substring]
cd /tmp && export TMPDIR=/tmp
CMD="$1"; KN="${2:-k_region_co}"
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  rm -rf /tmp/k2pmc
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/k2pmc -o p -- $CMD > /dev/null 2>&1
  python3 - "$KN" <<'PY'
import csv, glob, collections, sys
kn = sys.argv[1]
f = glob.glob("/tmp/k2pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if kn in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", max(len(v) for v in acc.values()) if acc else 0)
PY
done
