#!/usr/bin/env python
"""Golden chain of a whole synthetic genome through the SEQUENTIAL C oracle: BASELINE.json configs[3] (200 M cis PETs, 23
chromosomes, seeds 3000 + chromosome index, Hi-C mode -m 3 = eps 5000/7500/10000 x minPts 50/40/30/20 with the chained
cut of cLoops/pipe.py:241-281) -- or any other (n_total, cfg, eps list, minPts list).

Per step: the oracle's single_dbscan (cLoops/pipe.py:52-110 on cDBSCAN2's labels) on every chromosome, merged like
runDBSCAN (pipe.py:113-127: chromosomes without an inter-ligation box are skipped), the cut from the reference's own
estIntSelCutFrag (cLoops/ests.py:36-61, loaded from /root/reference when it is there, cloops_amd.ests -- pinned against
it by tests/test_pipe_host.py -- otherwise); at the end combineTwice over all steps and filterClusterByDis
(pipe.py:130-174).  Written: per step (eps, minPts, cut_in, n_in, n_inter, n_self, cut_out, frags), the final cut, and
per chromosome the number of surviving candidate boxes and an order-independent 64-bit checksum of them.

One persistent worker process per chromosome (the reference's parallel shape, pipe.py:117); ~4 min on 8 cores,
~70 s on the GPU box's host:

    python tests/golden/make_golden_synth200M_chain.py [n_total cfg out.json [eps,.. minPts,..]]
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from golden_util import box_checksum  # noqa: E402  (order-independent 64-bit checksum of box rows)


def _worker(conn, ci, length, n, seed):
    import oracle
    from cloops_amd.synth import synth_chrom
    X, Y = synth_chrom(n, length, seed)
    X = X.astype(np.int64)
    Y = Y.astype(np.int64)
    conn.send(("ready", ci))
    while True:
        msg = conn.recv()
        if msg is None:
            break
        eps, m, cut = msg
        t0 = time.perf_counter()
        r = oracle.single_dbscan("v2", X, Y, eps, m, cut)
        dt = time.perf_counter() - t0
        n_in = int(((Y - X) >= cut).sum())
        conn.send((np.asarray(r["dataI"], np.int64).reshape(-1, 4), len(r["dataS"]), n_in,
                   r["dis"].astype(np.int32), r["dss"].astype(np.int32), dt))
    conn.close()


def run_chain(n_total, cfg, eps, minPts, log=None):
    from cloops_amd.synth import chrom_sizes
    from cloops_amd import pipe
    try:
        import refload
        est = refload.ref_ests().estIntSelCutFrag if refload.available() else None
    except Exception:
        est = None
    if est is None:
        from cloops_amd.ests import estIntSelCutFrag as est
    sizes = chrom_sizes(n_total)
    ctx = mp.get_context("fork")
    procs = []
    for ci, (name, length, n) in enumerate(sizes):
        a, b = ctx.Pipe()
        p = ctx.Process(target=_worker, args=(b, ci, length, n, 1000 * cfg + ci), daemon=True)       # daemons: a failing parent takes them along
        p.start()
        procs.append((p, a))
    for p, a in procs:
        a.recv()
    steps, cut, cuts = [], 0, [0]
    per_chrom = [[] for _ in sizes]
    cpu_s = 0.0
    wall0 = time.perf_counter()
    for ep in eps:
        for m in minPts:
            t0 = time.perf_counter()
            for p, a in procs:
                a.send((ep, m, cut))
            nI = nS = n_in = 0
            dis, dss = [], []
            slowest = 0.0
            for ci, (p, a) in enumerate(procs):
                dI, ns, nin, di, ds, dt = a.recv()
                cpu_s += dt
                slowest = max(slowest, dt)
                n_in += nin
                if len(dI) == 0:                              # runDBSCAN, pipe.py:121-122
                    continue
                nI += len(dI)
                nS += ns
                dis.append(di)
                dss.append(ds)
                per_chrom[ci].append(dI)
            st = {"eps": ep, "minPts": m, "cut_in": int(cut), "n_in": n_in, "n_inter": nI, "n_self": nS,
                  "slowest_chromosome_s": round(slowest, 3), "wall_s": None}
            if nI:
                dis = np.concatenate(dis).astype(np.float64)
                dss = np.concatenate(dss).astype(np.float64)
                if len(dis) and len(dss):                     # pipe.py:256-259
                    c2, frags = est(dis, dss)
                    st["cut_out"], st["frags"] = int(c2), int(frags)
                    cuts.append(int(c2))
                    cut = int(c2)
            st["wall_s"] = round(time.perf_counter() - t0, 3)
            steps.append(st)
            if log:
                log("step %s" % json.dumps(st))
    for p, a in procs:
        a.send(None)
    for p, a in procs:
        p.join()
    pos = [c for c in cuts if c > 0]
    final_cut = int(min(pos))
    chroms = {}
    for ci, (name, length, n) in enumerate(sizes):
        rows = pipe._combine_steps(per_chrom[ci])
        rows = np.asarray(rows, np.int64).reshape(-1, 4)
        keep = ((rows[:, 2] + rows[:, 3]) // 2 - (rows[:, 0] + rows[:, 1]) // 2) >= final_cut      # pipe.py:130-143, py2 floor
        rows = rows[keep]
        chroms[name] = {"pets": n, "candidates": int(len(rows)), "checksum": box_checksum(rows)}
    return {"n_total": n_total, "cfg": cfg, "eps": list(eps), "minPts": list(minPts), "steps": steps, "cuts": cuts,
            "final_cut": final_cut, "candidates": sum(c["candidates"] for c in chroms.values()), "chromosomes": chroms,
            "oracle_cpu_s": round(cpu_s, 1), "oracle_wall_s": round(time.perf_counter() - wall0, 1), "workers": len(sizes),
            "host_cores": os.cpu_count()}


def main():
    n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 200000000
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(HERE, "synth200M_mode3_oracle_chain.json")
    eps = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [5000, 7500, 10000]
    minPts = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [50, 40, 30, 20]
    import oracle
    oracle.build()
    res = run_chain(n_total, cfg, eps, minPts, log=lambda s: sys.stderr.write(s + "\n"))
    res["generator"] = "tests/golden/make_golden_synth200M_chain.py (sequential C oracle, one process per chromosome)"
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    sys.stderr.write("wrote %s: cuts %s final %d candidates %d\n" % (out, res["cuts"], res["final_cut"], res["candidates"]))


if __name__ == "__main__":
    main()
