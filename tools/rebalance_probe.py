"""Developer tool: bench.py's timed sweep with (1) / without (0) the LPT re-deal of the sweep's chromosomes over the shared streams
(pipe.STREAMS.rebalance).  python tools/rebalance_probe.py 0"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cloops_amd.pipe as p
if not int(sys.argv[1]):
    p.STREAMS.rebalance = lambda chroms: None
import bench
sys.argv = ["bench.py", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-with-labels", "--no-secondary", "--proxy-ranks", "0"]
bench.main()
