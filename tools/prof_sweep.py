import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cloops_amd import pipe
from cloops_amd.synth import synth_genome
fs=[]
for name, X, Y in synth_genome(int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000, cfg=3):
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
pipe.runSweepFast(fs, [5000], [50], cut=0)
pr=cProfile.Profile(); pr.enable()
pipe.runSweepFast(fs, [5000, 7500, 10000], [50, 40, 30, 20], cut=0)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
