#!/bin/bash
# Per-kernel rocprofv3 stats (+ FETCH/WRITE PMC passes) for chr1 of the 200 M-PET genome at the mode-3 settings.
# usage (through gpurun): bash tools/profile_dense.sh <outdir-under-gpurun_out> [pmc]
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${1:-dense}
PMC=${2:-}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "5000 50 0" "7500 30 0" "7500 30 5000" "10000 20 5000"; do
  set -- $cfg
  tag=e$1_m$2_c$3
  rm -rf /tmp/pd_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd_$tag -o s -- python $REPO/tools/dense_run.py $1 $2 $3 6 > $OUT/run_$tag.txt 2>&1
  cp $(find /tmp/pd_$tag -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$tag.csv
  if [ -n "$PMC" ]; then
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pp_$tag
      rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pp_$tag -o p -- python $REPO/tools/dense_run.py $1 $2 $3 2 > /dev/null 2>&1
      cp $(find /tmp/pp_$tag -name "*counter_collection.csv" | head -1) /tmp/cc_${tag}_$ctr.csv
    done
    python - <<PY
import csv, collections, json, re
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("/tmp/cc_${tag}_%s.csv" % ctr)):
        if r["Counter_Name"] == ctr:
            acc[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[ctr + "_KB_avg"] = round(sum(v) / len(v), 1)
        out[k]["launches_per_run"] = len(v) / 2.0
json.dump(out, open("$OUT/pmc_fetch_write_$tag.json", "w"), indent=1)
PY
  fi
  python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats_$tag.csv")))
print("== $tag")
for r in rows[:16]:
    print("%-70s %5s %9.1f us %6s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  tail -2 $OUT/run_$tag.txt
done
