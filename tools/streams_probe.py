"""Developer tool: bench.py's timed sweep with pipe.SWEEP_STREAMS = argv[1] shared streams (1 .. 8).  python tools/streams_probe.py 3"""
import sys, os, json, io, contextlib
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import cloops_amd.pipe as p
p.SWEEP_STREAMS = int(sys.argv[1])
import bench
sys.argv = ["bench.py", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-with-labels", "--no-secondary", "--proxy-ranks", "0"]
bench.main()
