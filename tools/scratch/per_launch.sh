# per-launch durations of one kernel of a command.  usage: per_launch.sh "<cmd>" <kernel name substring>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl
rocprofv3 --kernel-trace --output-format csv -d /tmp/pl -o s -- $1 > /dev/null 2>&1
python3 - "$2" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob("/tmp/pl/**/*kernel_trace.csv", recursive=True)[0])))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if sys.argv[1] in r["Kernel_Name"]]
print(" ".join("%.1f" % x for x in d))
PY
