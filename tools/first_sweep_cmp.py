import csv, glob, sys, collections
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load(sys.argv[1]), load(sys.argv[2])     # 1 sweep, 3 sweeps
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    steady_c, steady_t = (cb - ca) / 2.0, (tb - ta) / 2.0
    rows.append((ta - steady_t, k[:60], ca, steady_c, ta, steady_t))
for r in sorted(rows, reverse=True)[:14]:
    print("extra %7.2f ms  %-60s calls first %5d steady %7.1f  ms first %7.2f steady %7.2f" % r)
print("total extra kernel ms:", sum(r[0] for r in rows))
