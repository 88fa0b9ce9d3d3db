"""Per-run totals of the roofline replay (chr1, the sweep's 12 runs in the sweep's order, region query re-used inside an eps): kernel time
and launches per run from the rocprofv3 kernel stats, HBM traffic per run and kernel from the FETCH_SIZE / WRITE_SIZE passes
(2 x FETCH + WRITE, profiles/README.md).   python tools/run_report.py [profiles/r4]"""
import csv
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r4")
ONE_OFF = ("k_dhist", "k_stats", "k_make_qkeys", "k_make_keys", "k_decode_sorted", "k_strip_table(", "k_init_pads", "k_nop", "k_export_table")
SORT = ("rocprim::ROCPRIM_400200_NS::detail::trampoline_kernel<rocprim::ROCPRIM_400200_NS::detail::radix", "k_decode_sp", "k_make_spkeys", "k_strip_table32")


def clean(n):
    return n.replace("void ", "")


rows = list(csv.DictReader(open(os.path.join(d, "k2_replay_reuse_kernel_stats.csv"))))
runs = [int(r["Calls"]) for r in rows if clean(r["Name"]).startswith("k_final_l")][0]
per = []
for r in rows:
    n = clean(r["Name"])
    if any(n.startswith(s) for s in ONE_OFF):
        continue
    per.append((float(r["TotalDurationNs"]) / 1e3 / runs, int(r["Calls"]) / runs, n[:72]))
per.sort(reverse=True)
print("kernel time per run: %.1f us in %.1f launches (%d runs; the layout sorts of the three eps are spread over their four runs each)" % (
    sum(p[0] for p in per), sum(p[1] for p in per), runs))
for p in per[:24]:
    print("  %8.1f us %5.2f launches  %s" % p)
j = json.load(open(os.path.join(d, "pmc_fetch_write_k2replay.json")))
runs2 = [v["launches_FETCH_SIZE"] for k, v in j.items() if k.startswith("k_final_l")][0]
det = []
for k, v in j.items():
    if any(k.startswith(s.rstrip("(")) for s in ONE_OFF):
        continue
    det.append(((2 * v["FETCH_SIZE_KB_sum"] + v["WRITE_SIZE_KB_sum"]) * 1024 / runs2 / 1e6, k[:72]))
det.sort(reverse=True)
tot = sum(x[0] for x in det)
sort = sum(x[0] for x in det if any(x[1].startswith(s[:40]) for s in SORT) or "radix" in x[1] or "onesweep" in x[1])
print("HBM traffic per run: %.0f MB (%d runs), of which layout sorts %.0f MB" % (tot, runs2, sort))
for x in det[:18]:
    print("  %8.1f MB  %s" % x)
