#!/bin/bash
# developer tool: the steady-state sweep as a timeline -- per hardware queue busy time, how many kernels run side by side, the sum of
# the in-sweep kernel durations by kernel.  usage (through gpurun): bash tools/sweep_timeline.sh [n_total] [mode]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gb
rocprofv3 --kernel-trace --output-format csv -d /tmp/gb -o t -- python $GRAFT_REPO_ROOT/tools/sweep_bench.py ${1:-200e6} ${2:-3} --fast-only > /tmp/gb_out.txt 2>&1
grep rep /tmp/gb_out.txt
python3 - <<'PY'
import csv, glob, re, collections
f = glob.glob("/tmp/gb/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:48], r.get("Queue_Id", "?")) for r in rows)
rep = [float(m) for m in re.findall(r"chained cut\) ([0-9.]+) s", open("/tmp/gb_out.txt").read())]
t_end = iv[-1][1]
span0 = t_end - int(rep[-1] * 1e9)
sel = [x for x in iv if x[0] >= span0]
tot = sel[-1][1] - sel[0][0]
ev = []
for a, b, _, _ in sel:
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
conc = collections.Counter(); cur = 0; last = ev[0][0]
for t, d in ev:
    conc[cur] += t - last; last = t; cur += d
print("last sweep: window %.1f ms, %d kernels, sum of durations %.1f ms (%.2f x the window)" % (tot / 1e6, len(sel), sum(b - a for a, b, _, _ in sel) / 1e6, sum(b - a for a, b, _, _ in sel) / tot))
print("kernels running side by side (share of the window):", {k: "%.1f %%" % (100.0 * v / tot) for k, v in sorted(conc.items())})
byq = collections.defaultdict(int)
for a, b, _, q in sel:
    byq[q] += b - a
print("busy per hardware queue (ms):", {q: round(v / 1e6, 1) for q, v in sorted(byq.items())})
byk = collections.defaultdict(lambda: [0, 0])
for a, b, k, _ in sel:
    byk[k][0] += b - a; byk[k][1] += 1
for k, (t, n) in sorted(byk.items(), key=lambda kv: -kv[1][0])[:24]:
    print("%-50s %6d launches %8.1f ms total %8.1f us avg" % (k, n, t / 1e6, t / n / 1e3))
PY
