"""Developer tool: per-run kernel durations of the roofline replay from a rocprofv3 --kernel-trace CSV: the last pass of 12 runs,
one column per run.  python tools/per_run_trace.py <dir with *_kernel_trace.csv> [marker kernel prefix]"""
import csv, glob, re, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_final_l"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def clean(n):
    return re.sub(r"\(.*", "", n).replace("void ", "")[:34]
runs, cur = [], []
for r in rows:
    n = clean(r["Kernel_Name"])
    cur.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    if n.startswith(marker):
        runs.append(cur); cur = []
runs = runs[-12:]
names = []
for run in runs:
    for n, _ in run:
        if n not in names:
            names.append(n)
print("%-34s" % "kernel" + "".join("%7d" % k for k in range(len(runs))))
tot = [0.0] * len(runs)
for n in names:
    vals = [sum(t for m, t in run if m == n) for run in runs]
    if max(vals) < 3.0:
        continue
    for k, v in enumerate(vals):
        tot[k] += v
    print("%-34s" % n + "".join("%7.1f" % v for v in vals))
print("%-34s" % "sum (listed)" + "".join("%7.0f" % v for v in tot))
