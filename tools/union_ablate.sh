#!/bin/bash
# developer tool: where k_union_cores' time goes -- the developer library leaves parts of it out (CLOOPS_DBG bits; results invalid).
# usage (GPU box): bash tools/union_ablate.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export CLOOPS_DEVEL_LIB=1
export CLOOPS_SKIP=141        # nothing behind k_union_cores runs on its (invalid) output
for dbg in 0 16777216 33554432 67108864; do
  echo -n "CLOOPS_DBG=$dbg (16777216 staging + core list only | 33554432 searches, no candidates | 67108864 no union-find step): "
  CLOOPS_DBG=$dbg timeout 120 bash $R/tools/kstats.sh "python $R/tools/k2_replay.py 1" 60 | grep -E "k_union" | awk '{print $(NF-2), $(NF-1)}'
done
