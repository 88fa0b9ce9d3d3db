#!/bin/bash
# developer tool: A/B of the shipped library (A) against cloops_amd/libcloops_hip_devel.so (B) on ONE box, alternating runs of
# bench.py (the wall clock of a sweep varies by a few per cent between boxes).  usage (through gpurun): bash tools/ab_bench.sh [reps]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for k in $(seq 1 ${1:-3}); do
  for d in 0 1; do
    CLOOPS_DEVEL_LIB=$d python $REPO/bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('lib', '$d', 'ms_per_step %.1f' % j['ms_per_step'])"
  done
done
