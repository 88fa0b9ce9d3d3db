#!/bin/bash
# developer tool: tools/queue_probe.py over GPU_MAX_HW_QUEUES, with the sweep's shared streams (pipe.SWEEP_STREAMS) and with one
# stream per chromosome.  usage (through gpurun): bash tools/queue_matrix.sh [n_total]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-200000000}
for q in 2 3 4 8; do
  GPU_MAX_HW_QUEUES=$q python $R/tools/queue_probe.py "queues=$q streams=3" $N 3 2>&1 | tail -1
done
python $R/tools/queue_probe.py "queues=default streams=3" $N 3 2>&1 | tail -1
python $R/tools/queue_probe.py "queues=default streams=2" $N 2 2>&1 | tail -1
python $R/tools/queue_probe.py "queues=default streams=4" $N 4 2>&1 | tail -1
for q in 3 4 8; do
  GPU_MAX_HW_QUEUES=$q python $R/tools/queue_probe.py "queues=$q stream/chrom" $N 0 2>&1 | tail -1
done
