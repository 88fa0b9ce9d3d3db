#!/bin/bash
# developer tool: where k_chain_flags' time goes -- the developer library leaves parts of it out (CLOOPS_DBG bits; results invalid).
# usage (GPU box): bash tools/chain_ablate.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export CLOOPS_DEVEL_LIB=1
export CLOOPS_SKIP=207        # nothing behind k_chain_flags runs on its (invalid) output
for dbg in 0 134217728 268435456 536870912 1073741824 805306368; do
  echo -n "CLOOPS_DBG=$dbg (134217728 staging only | 268435456 no cell heads | 536870912 no own-strip walks | 1073741824 no LDS atomicMin / row load): "
  CLOOPS_DBG=$dbg timeout 120 bash $R/tools/kstats.sh "python $R/tools/k2_replay.py 1" 60 | grep -E "k_chain_flags" | awk '{print $(NF-2), $(NF-1)}'
done
