"""where does the host spend the time between two steps of a sweep?  (monkey-patches timers into pipe._sweep_fast's helpers)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cloops_amd import pipe, api
from cloops_amd.synth import synth_genome
fs = []
for name, X, Y in synth_genome(200000000, cfg=3):
    fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y))
ev = []
orig_step = api.Chromosome.step_async
orig_wait = api.Chromosome.wait
orig_res = api.Chromosome.step_result
def step_async(self, *a, **k):
    t0 = time.perf_counter(); r = orig_step(self, *a, **k); ev.append(("enq", t0, time.perf_counter())); return r
def wait(self, *a, **k):
    t0 = time.perf_counter(); r = orig_wait(self, *a, **k); ev.append(("wait", t0, time.perf_counter())); return r
def step_result(self, *a, **k):
    t0 = time.perf_counter(); r = orig_res(self, *a, **k); ev.append(("res", t0, time.perf_counter())); return r
api.Chromosome.step_async = step_async; api.Chromosome.wait = wait; api.Chromosome.step_result = step_result
for rep in range(3):
    ev.clear()
    t0 = time.perf_counter()
    pipe.runSweepFast(fs, [5000, 7500, 10000], [50, 40, 30, 20], cut=0)
    dt = time.perf_counter() - t0
enq = sorted([e for e in ev if e[0] == "enq"], key=lambda e: e[1])
waits = [e for e in ev if e[0] == "wait"]
ress = [e for e in ev if e[0] == "res"]
n = len(fs)
print("sweep %.3f s; enqueue calls %d (mean %.0f us); step_result mean %.0f us" % (dt, len(enq), 1e6 * np.mean([e[2] - e[1] for e in enq]), 1e6 * np.mean([e[2] - e[1] for e in ress])))
for s in range(1, 12):
    first_enq = enq[s * n]
    prev_waits = [w for w in waits if w[2] <= first_enq[1]]
    last_wait_end = max(w[2] for w in prev_waits)
    last_res_end = max(r[2] for r in ress if r[2] <= first_enq[1])
    last_enq_end = enq[s * n + n - 1][2]
    print("step %2d: last wait returned -> last step_result done %4.0f us -> first enqueue starts %4.0f us -> first enqueue done %4.0f us -> all %d enqueued %5.0f us" % (
        s, 1e6 * (last_res_end - last_wait_end), 1e6 * (first_enq[1] - last_res_end), 1e6 * (first_enq[2] - first_enq[1]), n, 1e6 * (last_enq_end - first_enq[1])))
