#!/bin/bash
# developer tool: rocprofv3 per-kernel averages of a command.  usage: tools/kstats.sh "<cmd>" [rows]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- $1 > /dev/null 2>&1
python3 - "${2:-32}" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.1f us" % (tot / 1e3))
for r in rows[:int(sys.argv[1])]:
    print("%-86s %5s %9.1f us %6s%%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
