#!/bin/bash
# Everything profiles/rN/ holds, regenerated on the GPU box in one gpurun call (every step under its own timeout) -> gpurun_out/round/.
# usage: bash tools/round_artifacts.sh        then copy gpurun_out/round/* and gpurun_out/prof/* into profiles/rN/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/round
mkdir -p $O $R/gpurun_out/prof
cd $R
timeout 400 python bench.py > $O/bench_final.json 2> $O/bench_final.err
timeout 600 bash tools/profile_bench.sh > $R/gpurun_out/prof/profile_bench.log 2>&1
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/pst && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pst -o t -- python $R/tools/one_chrom_steps.py 0 > /dev/null 2>&1; python $R/tools/per_run_trace.py /tmp/pst k7_reduce_parts > $O/per_step_trace.txt 2>&1 )
( cd /tmp && rm -rf /tmp/prt && CLOOPS_REPLAY_ONLY=reuse timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prt -o t -- python $R/tools/k2_replay.py 1 > /dev/null 2>&1; python $R/tools/per_run_trace.py /tmp/prt > $O/per_run_trace.txt 2>&1 )
CLOOPS_REPLAY_ONLY=reuse timeout 300 bash tools/pmc_all.sh "python $R/tools/k2_replay.py 1" > $O/pmc_sq_counters_k2replay.txt 2>&1
( for a in "40e6 4" "200e6 3" "500e6 5"; do timeout 400 python tools/sweep_bench.py $a --fast-only 2>&1 | tail -3; done ) > $O/sweeps_one_gpu.txt 2>&1
( for s in 101 102 103; do timeout 300 python tools/fuzz_gpu.py $s 400 2>&1 | tail -1; done
  for s in 101 102; do timeout 300 python tools/fuzz_lists.py $s 60 2>&1 | tail -1; done
  for s in 101 102 103 104; do CLOOPS_DEVEL_LIB=1 CLOOPS_DBG=4096 timeout 300 python tools/fuzz_k2_keys.py $s 40 2>&1 | tail -2; done
  for s in 101 102; do timeout 300 python tools/fuzz_layouts.py $s 40 2>&1 | tail -1; done
  timeout 200 python tools/fuzz_weighted.py 101 200 2>&1 | tail -1
  timeout 300 python tools/lists_check.py 2>&1 | tail -1 ) > $O/fuzz_final_code.txt 2>&1
CLOOPS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-with-labels --proxy-ranks 0 --no-secondary > $O/bench_world1_over_rccl.json 2> /dev/null
tail -3 $O/fuzz_final_code.txt; head -c 300 $O/bench_final.json
