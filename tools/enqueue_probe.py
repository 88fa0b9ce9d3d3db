"""Developer tool: bench.py's timed sweep with the step's runs enqueued by the host pool (pipe.PARALLEL_ENQUEUE = argv[1], 0 / 1) --
is the sweep bound by the ONE host thread that enqueues ~6 700 kernels per sweep?  python tools/enqueue_probe.py 1"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cloops_amd.pipe as p
p.PARALLEL_ENQUEUE = bool(int(sys.argv[1]))
import bench
sys.argv = ["bench.py", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-with-labels", "--no-secondary", "--proxy-ranks", "0"]
bench.main()
