# developer A/B (needs `python -m cloops_amd.build --devel`): label copies on one copy stream per shared stream (0) vs inside the compute streams (1)
export CLOOPS_DEVEL_LIB=1
for k in 1 2; do
for m in 0 1; do
  if [ $m = 1 ]; then export CLOOPS_COPY_IN_STREAM=1; else unset CLOOPS_COPY_IN_STREAM; fi
  timeout 300 python bench.py --steps 2 --no-cpu-baseline --no-secondary --proxy-ranks 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('copy_in_stream', '$m', 'ms_per_step %.1f' % j['ms_per_step'], 'with_labels %.4f' % j['with_labels']['sweep_wall_s'])"
done; done
