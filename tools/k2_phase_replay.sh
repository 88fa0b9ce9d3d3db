#!/bin/bash
# developer tool: the k_region_core launches of the roofline replay (chr1, announced lists) stopped after each phase
# (devel library, CLOOPS_DBG bits: 32 staging, 64 phase 0, 128 phase 1, 512 searches of phase 2, 2048 no phase 3, 0 all).
# Prints the duration of every k_region_core launch of the replay in launch order.  usage: tools/k2_phase_replay.sh [dbg ...]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export CLOOPS_DEVEL_LIB=1 CLOOPS_REPLAY_ONLY=reuse
python -m cloops_amd.build --devel > /dev/null 2>&1 || (cd $REPO && python -m cloops_amd.build --devel > /dev/null 2>&1)
for DBG in ${@:-32 64 128 512 2048 0}; do
  rm -rf /tmp/k2ph
  CLOOPS_DBG=$DBG timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/k2ph -o p -- python $REPO/tools/k2_replay.py 1 > /tmp/k2ph.log 2>&1
  python3 - "$DBG" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/k2ph/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_region_" in r["Kernel_Name"] or "k_band" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
core = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_region_" in r["Kernel_Name"]]
band = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_band" in r["Kernel_Name"]]
print("dbg=%-5s k_region_core us: %s | k_band avg %.1f (%d)" % (sys.argv[1], " ".join("%.1f" % x for x in core), sum(band) / max(1, len(band)), len(band)))
PY
done
