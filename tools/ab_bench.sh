#!/bin/bash
# developer tool: A/B of the shipped library (A) against cloops_amd/libcloops_hip_devel.so (B; a devel build or the library of an
# earlier commit copied there) on ONE box, alternating runs of bench.py's timed sweeps (the wall clock of a sweep varies by a few
# per cent between boxes).  usage (through gpurun): bash tools/ab_bench.sh [reps]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for k in $(seq 1 ${1:-3}); do
  for d in 0 1; do
    CLOOPS_DEVEL_LIB=$d python $REPO/bench.py --steps 4 --no-cpu-baseline --no-secondary --no-with-labels --proxy-ranks 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('lib', '$d', 'ms_per_step %.1f' % j['ms_per_step'], 'first %.3f' % j['first_sweep_s'])"
  done
done
