#!/bin/bash
# developer tool: where k_border's time goes -- the developer library leaves parts of it out (CLOOPS_DBG bits; results invalid).
# usage (GPU box): bash tools/border_ablate.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export CLOOPS_DEVEL_LIB=1
export CLOOPS_SKIP=136        # nothing behind k_border runs on its (invalid) output
for dbg in 0 65536 131072 262144 524288 1048576 2097152 4194304 8388608; do
  echo -n "CLOOPS_DBG=$dbg (65536 staging only | 131072 no walkers | 262144 no own strip | 524288 no strip s-1 | 1048576 no strip s+1 | 2097152 first four candidates only | 4194304 no walk starting outside the window | 8388608 no global continuation): "
  CLOOPS_DBG=$dbg timeout 120 bash $R/tools/kstats.sh "python $R/tools/k2_replay.py 1" 60 | grep -E "k_border" | awk '{print $(NF-2), $(NF-1)}'
done
