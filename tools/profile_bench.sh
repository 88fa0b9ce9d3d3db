#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box (run through gpurun), into gpurun_out/prof/:
#   bench_kernel_stats.csv       rocprofv3 --kernel-trace --stats of `python bench.py` (the command the driver runs)
#   k2_replay_{reuse,full}_kernel_stats.csv   the same for the roofline leg alone (tools/k2_replay.py: chr1, the sweep's 12 runs in the
#                                sweep's order), with the region query re-used inside an eps / with every run doing its own; k2_from_rocprof.json
#   pmc_fetch_write_k2replay.json  FETCH_SIZE / WRITE_SIZE per kernel over that replay (separate --pmc passes)
#   k2_traffic.json              HBM bytes of the region query per RUN, amortised like roofline.achieved -> bench.py roofline.traffic
# usage: bash tools/profile_bench.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_stats /tmp/pb_k2 /tmp/pb_fetch /tmp/pb_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_stats -o b -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/pb_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
# the roofline leg alone, once with the region query re-used inside an eps (what a sweep executes) and once with every run doing its own
for kind in reuse full; do
  rm -rf /tmp/pb_k2
  CLOOPS_REPLAY_ONLY=$kind rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_k2 -o k -- python $REPO/tools/k2_replay.py 3 > $OUT/k2_replay_$kind.txt 2>&1
  cp $(find /tmp/pb_k2 -name "*kernel_stats.csv" | head -1) $OUT/k2_replay_${kind}_kernel_stats.csv
done
python $REPO/tools/k2_replay.py 3 > $OUT/k2_replay.txt 2>&1
# counters: the re-using passes only, one pass
CLOOPS_REPLAY_ONLY=reuse rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pb_fetch -o f -- python $REPO/tools/k2_replay.py 1 > /dev/null 2>&1
CLOOPS_REPLAY_ONLY=reuse rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pb_write -o w -- python $REPO/tools/k2_replay.py 1 > /dev/null 2>&1
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
              "FETCH_SIZE_KB_sum": round(sum(fe[k]), 1), "WRITE_SIZE_KB_sum": round(sum(wr[k]), 1),
              "launches_FETCH_SIZE": len(fe[k]), "launches_WRITE_SIZE": len(wr[k])}
json.dump(out, open("$OUT/pmc_fetch_write_k2replay.json", "w"), indent=1)
# the region query of a sweep's 12 runs on chr1 = the k_region_core launches (first run of every eps) + what k_cut_copy<true> (cut compaction
# + the query on the cut band + nothing else) moves beyond a plain k_cut_copy<false>; 2 x warm-up and timed pass -> per run
def kb(k, what):
    return out[k][what] if k in out else 0.0
k2s = [k for k in out if "k_region_core" in k or "k_region_keys" in k]
runs = 0
for k in out:
    if k.startswith("k_final_l"):
        runs = out[k]["launches_FETCH_SIZE"]
hbm = lambda k: 2 * kb(k, "FETCH_SIZE_KB_sum") + kb(k, "WRITE_SIZE_KB_sum")
ct = [k for k in out if k.startswith("k_cut_copy<true>")]
cf = [k for k in out if k.startswith("k_cut_copy<false>")]
kb_band = [k for k in out if k.startswith("k_band")]     # traversal level 4: the band query is a kernel of its own, nothing is copied
plain = (hbm(cf[0]) / out[cf[0]]["launches_FETCH_SIZE"]) if cf else 0.0
band_extra = sum(hbm(k) - plain * out[k]["launches_FETCH_SIZE"] for k in ct) + sum(hbm(k) for k in kb_band)
total_kb = sum(hbm(k) for k in k2s) + max(band_extra, 0.0)
t = {"workload": "chr1 (16.4 M PETs) of synthetic-200M-23chr, the mode-3 sweep's 12 runs in the sweep's order, region query re-used inside an eps",
     "kernels": k2s + ct + kb_band, "runs": runs,
     "region_core_KB_per_launch": {k: round(hbm(k) / max(1, out[k]["launches_FETCH_SIZE"]), 1) for k in k2s},
     "band_query_KB_per_launch": round(band_extra / max(1, sum(out[k]["launches_FETCH_SIZE"] for k in ct + kb_band)), 1) if (ct or kb_band) else None,
     "correction": "gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is 1:1",
     "hbm_bytes_per_launch": int(round(total_kb * 1024 / max(1, runs))),
     "hbm_bytes_per_launch_note": "per RUN of the sweep (amortised like roofline.achieved): all region-query traffic of the replay / its runs",
     "source": "tools/profile_bench.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over tools/k2_replay.py)"}
json.dump(t, open("$OUT/k2_traffic.json", "w"), indent=1)
print(json.dumps(t))
# the amortised region-query time from the rocprofv3 kernel durations alone (cross-check of bench.py's HIP-event figure): per pass of 12 runs
# = all k_region_core launches of the re-using passes + what k_cut_copy<true> / k_cut_strips take there beyond the plain compaction of the same runs
def stats(name):
    return {re.sub(r"\(.*", "", r["Name"]).replace("void ", ""): (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open("$OUT/" + name))}
ru, fu = stats("k2_replay_reuse_kernel_stats.csv"), stats("k2_replay_full_kernel_stats.csv")
passes = [v for k, v in ru.items() if k.startswith("k_final_l")][0][0] // 12
core = sum(c * a for k, (c, a) in ru.items() if k.startswith(("k_region_core", "k_region_keys")))
plain = fu["k_cut_copy<false>"][1] if "k_cut_copy<false>" in fu else 0.0
if "k_band" in ru:                                       # level 4: the band query of every run under a cut, a kernel of its own
    carry = ru["k_band"][0] * ru["k_band"][1]
else:
    carry = ru["k_cut_copy<true>"][0] * (ru["k_cut_copy<true>"][1] - plain) + ru["k_cut_strips"][0] * (ru["k_cut_strips"][1] - fu["k_cut_strips"][1])
full = sum(c * a for k, (c, a) in fu.items() if k.startswith(("k_region_core", "k_region_keys")))
x = {"passes": passes, "runs_per_pass": 12, "region_core_us_per_pass": round(core / passes, 1), "carry_us_per_pass": round(carry / passes, 1),
     "amortised_us_per_run": round((core + carry) / passes / 12, 2), "full_query_us_per_run": round(full / passes / 12, 2),
     "kernels_reuse": {k: ru[k] for k in ru if k.startswith(("k_region_core", "k_region_keys", "k_cut_copy", "k_cut_strips", "k_band"))},
     "per_launch_us": {k: ru[k][1] for k in ru if k.startswith(("k_region_core", "k_region_keys"))},
     "kernels_full": {k: fu[k] for k in fu if k.startswith(("k_region_core", "k_region_keys", "k_cut_copy", "k_cut_strips"))},
     "source": "k2_replay_reuse_kernel_stats.csv / k2_replay_full_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/k2_replay.py 3 with CLOOPS_REPLAY_ONLY=reuse / full; the first of the four passes is the warm-up)"}
json.dump(x, open("$OUT/k2_from_rocprof.json", "w"), indent=1)
print(json.dumps(x))
for name in ("bench_kernel_stats.csv", "k2_replay_reuse_kernel_stats.csv"):
    rows = list(csv.DictReader(open("$OUT/" + name)))
    print("==", name)
    for r in rows[:14]:
        print(r["Name"][:64], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
tail -14 $OUT/k2_replay.txt; cat $OUT/bench_under_rocprof.json | cut -c1-400
