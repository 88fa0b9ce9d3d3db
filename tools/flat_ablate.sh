#!/bin/bash
# developer tool: where the time of k_flatten / k_final_labels goes -- the developer library leaves parts out (CLOOPS_DBG2 bits;
# results invalid).  usage (GPU box): bash tools/flat_ablate.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export CLOOPS_DEVEL_LIB=1
for dbg in 0 1 2 4; do
  echo -n "k_flatten CLOOPS_DBG2=$dbg (1 loads only | 2 no forest walk | 4 no root list / aggregation): "
  CLOOPS_SKIP=137 CLOOPS_DBG2=$dbg timeout 120 bash $R/tools/kstats.sh "python $R/tools/k2_replay.py 1" 60 | grep -E "k_flatten" | awk '{print $(NF-2), $(NF-1)}'
done
for dbg in 0 16 32; do
  echo -n "k_final_labels CLOOPS_DBG2=$dbg (16 owner -> label only | 32 no cluster table): "
  CLOOPS_DBG2=$dbg timeout 120 bash $R/tools/kstats.sh "python $R/tools/k2_replay.py 1" 60 | grep -E "k_final_labels" | awk '{print $(NF-2), $(NF-1)}'
done
