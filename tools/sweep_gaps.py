"""Developer tool: how much of a sweep's wall clock has NO kernel running on the device (host-bound time), from a rocprofv3
--kernel-trace csv of bench.py:   python tools/sweep_gaps.py /tmp/dir   (looks at the last 0.25 s of kernel activity)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
iv = []
for r in csv.DictReader(open(f)):
    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
iv.sort()
end = iv[-1][1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
lo = end - int(win * 1e9)
iv = [(max(a, lo), b) for a, b in iv if b > lo]
busy, cur_a, cur_b, gaps = 0, None, None, []
for a, b in iv:
    if cur_a is None: cur_a, cur_b = a, b; continue
    if a <= cur_b: cur_b = max(cur_b, b)
    else:
        busy += cur_b - cur_a; gaps.append(a - cur_b); cur_a, cur_b = a, b
busy += cur_b - cur_a
wall = end - lo
ksum = sum(b - a for a, b in iv)
gaps.sort(reverse=True)
print("window %.3f s: %d kernels, device busy %.1f %% of the wall clock, sum of kernel durations %.2f x the wall clock" % (wall / 1e9, len(iv), 100.0 * busy / wall, ksum / wall))
print("idle: %.2f ms in %d gaps; the longest: %s us" % ((wall - busy) / 1e6, len(gaps), ", ".join("%.0f" % (g / 1e3) for g in gaps[:12])))
big = [g for g in gaps if g > 20000]
print("gaps > 20 us: %d, %.2f ms" % (len(big), sum(big) / 1e6))
