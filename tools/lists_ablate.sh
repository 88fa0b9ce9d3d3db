#!/bin/bash
# developer tool (devel library): what the list kernels of a chr1 run spend where -- rocprofv3 averages of the roofline
# replay with parts of k_classify / k_make_lists / k_border_w switched off (results invalid).  bash tools/lists_ablate.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for d in ${ABL:-256 512 1024 2048 4096}; do
  echo "== CLOOPS_DBG2=$d"
  CLOOPS_DEVEL_LIB=1 CLOOPS_DBG2=$d CLOOPS_REPLAY_ONLY=reuse timeout 150 bash $R/tools/kstats.sh "timeout 120 python $R/tools/k2_replay.py 2" 12 2>&1 | grep -E "k_classify|k_make_lists|k_border_w|k_union_c|k_flatten_c|k_final_lists|k_chain_c|total"
done
