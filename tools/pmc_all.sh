#!/bin/bash
# developer tool: SQ counters per kernel (averages per launch) for a command.  usage: tools/pmc_all.sh "<cmd>" [min_us]
cd /tmp && export TMPDIR=/tmp
CMD="$1"
rm -rf /tmp/pa1 /tmp/pa2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pa1 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pa2 -o p -- $CMD > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/pa1", "/tmp/pa2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:34]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES"]
print("%-34s %5s " % ("kernel", "n") + " ".join("%11s" % c.replace("SQ_", "")[:11] for c in cols) + "  VALU/wave busy_us")
rows = []
for k, v in acc.items():
    m = {c: (sum(v[c]) / len(v[c]) if v.get(c) else 0.0) for c in cols}
    rows.append((m["SQ_BUSY_CYCLES"], k, len(v.get("SQ_WAVES", [])), m))
for busy, k, n, m in sorted(rows, reverse=True)[:24]:
    print("%-34s %5d " % (k, n) + " ".join("%11.0f" % m[c] for c in cols) + "  %8.0f %7.1f" % (m["SQ_INSTS_VALU"] / max(m["SQ_WAVES"], 1), busy / 32 / 2400.0))
PY
