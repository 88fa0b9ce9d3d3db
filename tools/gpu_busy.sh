#!/bin/bash
# developer tool: how busy is the GPU during the steady-state sweep?  union of kernel intervals / wall
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gb
rocprofv3 --kernel-trace --output-format csv -d /tmp/gb -o t -- python $GRAFT_REPO_ROOT/tools/sweep_bench.py ${1:-200e6} ${2:-3} --fast-only > /tmp/gb_out.txt 2>&1
grep rep /tmp/gb_out.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/gb/**/*kernel_trace.csv", recursive=True)[0]
iv = []
for r in csv.DictReader(open(f)):
    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
iv.sort()
t_end = iv[-1][1]
# steady-state sweep = the last 45 % of the trace time span after the first long gap ... simpler: take the last sweep by time: find the
# biggest gap between consecutive kernels in the second half (host work between the two timed sweeps is small), so use the last 0.6 s
import re
rep = [float(m) for m in re.findall(r"chained cut\) ([0-9.]+) s", open("/tmp/gb_out.txt").read())]
span0 = t_end - int(rep[-1] * 1e9)        # the last (steady-state) sweep
sel = [(a, b) for a, b, _ in iv if a >= span0]
busy = 0; cur_a, cur_b = sel[0]
gaps = []
for a, b in sel[1:]:
    if a > cur_b:
        busy += cur_b - cur_a; gaps.append(a - cur_b); cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
tot = sel[-1][1] - sel[0][0]
print("window %.1f ms: GPU busy (union of kernels) %.1f ms = %.0f %%; sum of kernel durations %.1f ms; %d kernels" % (tot / 1e6, busy / 1e6, 100.0 * busy / tot, sum(b - a for a, b in sel) / 1e6, len(sel)))
gaps.sort(reverse=True)
print("largest idle gaps (ms):", [round(g / 1e6, 2) for g in gaps[:16]], " gaps > 0.2 ms:", sum(1 for g in gaps if g > 2e5), "total idle in gaps > 50 us: %.1f ms" % (sum(g for g in gaps if g > 5e4) / 1e6))
PY
