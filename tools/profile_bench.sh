#!/bin/bash
# Regenerates the judged profile artefacts of bench.py on the GPU box (run through gpurun):
#   gpurun_out/prof/bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats
#   gpurun_out/prof/pmc_fetch_write.json     per-kernel FETCH_SIZE / WRITE_SIZE averages (separate --pmc passes)
#   gpurun_out/prof/k2_traffic.json          HBM bytes per K2 launch (2*FETCH + WRITE, see profiles/README.md)
# usage: bash tools/profile_bench.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_stats /tmp/pb_fetch /tmp/pb_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_stats -o b -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/pb_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pb_fetch -o f -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pb_write -o w -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, glob, json, collections, re
def load(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
                acc[k].append(float(r["Counter_Value"]))
    return acc
fe, wr = load("/tmp/pb_fetch", "FETCH_SIZE"), load("/tmp/pb_write", "WRITE_SIZE")
out = {}
for k in sorted(set(fe) | set(wr)):
    out[k] = {"FETCH_SIZE_KB_avg": round(sum(fe[k]) / max(1, len(fe[k])), 1), "WRITE_SIZE_KB_avg": round(sum(wr[k]) / max(1, len(wr[k])), 1),
              "launches_FETCH_SIZE": len(fe[k]), "launches_WRITE_SIZE": len(wr[k])}
json.dump(out, open("$OUT/pmc_fetch_write.json", "w"), indent=1)
k2 = [k for k in out if "k_region_count" in k][0]
mk = out.get("k_make_keys", {})
t = {"workload": "synthetic-5M-chr1-eps2000-minPts5", "kernel": k2, "FETCH_SIZE_KB": out[k2]["FETCH_SIZE_KB_avg"],
     "WRITE_SIZE_KB": out[k2]["WRITE_SIZE_KB_avg"],
     "correction": "gfx950 FETCH_SIZE reports 1/2 of the bytes read (calibration: k_make_keys reads exactly 40.0 MB and reports %s KB); WRITE_SIZE is 1:1" % mk.get("FETCH_SIZE_KB_avg"),
     "hbm_bytes_per_launch": int(round((2 * out[k2]["FETCH_SIZE_KB_avg"] + out[k2]["WRITE_SIZE_KB_avg"]) * 1024)),
     "source": "tools/profile_bench.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)"}
json.dump(t, open("$OUT/k2_traffic.json", "w"), indent=1)
print(json.dumps(t))
rows = list(csv.DictReader(open("$OUT/bench_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:64], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
cat $OUT/bench_under_rocprof.json | cut -c1-300
