#!/bin/bash
# LDS counters of a command's kernels (developer tool): bank conflicts vs active cycles
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pl -o p -- $1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:34]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "k_region_core" in k or "k_border" in k or "k_union" in k:
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
