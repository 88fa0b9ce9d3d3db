#!/usr/bin/env python
"""bench.py -- PETs clustered / s and sweep wall-clock on MI355X for the cDBSCAN hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (default): BASELINE.json configs[3] -- the synthetic genome of SURVEY.md 8d, 200 M cis PETs over the
23 hg38 chromosomes, Hi-C mode `-m 3` = eps [5000, 7500, 10000] x minPts [50, 40, 30, 20]
(cLoops/pipe.py:337-340) with the CHAINED distance cut of cLoops/pipe.py:247-275, variant cDBSCAN2 (the production
class, cLoops/pipe.py:42), through `cloops_amd.pipe.runSweepFast`.

A "step" is one pass of the hot path over the batch: ONE whole 12-run sweep over all 23 chromosomes, from "X,Y
resident in HBM" to "deduplicated, distance-filtered candidate-loop table on the host" (cluster tables cross
PCIe every run; labels stay on the device -- the sweep never needs them).  `value` = PETs that entered DBSCAN,
summed over the 12 runs and the K timed sweeps, / wall time; `sweep_wall_s` = ms_per_step / 1000 is the second
half of the metric.  Every timed sweep builds the layout of each of its three eps (a per-strip merge of the handle's fine
layout: the sorted order does not depend on minPts, and a cut only removes a PREFIX of every strip, so the four minPts
runs of an eps work on that one layout by index -- nothing is copied), one region query per eps plus the cut band of
every later run, and components, borders, cluster table and distance statistics for every run.  What a timed sweep
re-uses from the warm-up is the q index and the fine layout (they depend on the rows only): a real pipe() sweeps a
dataset once (cLoops/pipe.py:247-275), so the same sweep with every derived order dropped first (cl_chrom_drop_indexes;
allocations kept) is reported as top-level `cold_sweep_s`, the first sweep of the process (allocations too) as
`first_sweep_s`, and the sweep that lands labels + tables on the host every run (SURVEY.md 8d(1)'s end point) as
top-level `with_labels_sweep_s`.

With N > 1 (`--gpus N` spawns N ranks through torch.distributed.run -- only the launcher -- when not already launched by
it) the 23 chromosomes are LPT-sharded over the ranks (cloops_amd.dist.lpt_assign; chromosomes are independent units,
cLoops/pipe.py:117), every run exchanges ~48 KB of statistics (all-reduce: the chained cut is a genome-wide estimate) and
the final candidate tables are gathered once per sweep -- the path's only exchanges, both over RCCL through
libcloops_comm.so (cloops_amd/comm.py; no PyTorch in the GPU processes).  Total work is fixed (the same 200 M PETs):
"scaling": "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline"      K2 region-query kernel on chr1 (16.4 M PETs) at the sweep's own 12 (eps, minPts, cut) settings, timed
                  with HIP events on the library's stream right after the timed sweeps, chr1 ALONE on the GPU: inside a
                  sweep the kernels of the 23 chromosomes overlap on the device, so an event bracket there measures
                  contention, not the kernel (that figure is kept as `in_sweep_avg_launch_ms`)
  "secondary_5M"  BASELINE.json configs[1] (5 M PETs, one chromosome, eps 2000, minPts 5), the round-1 headline
  "cpu_baseline"  the CPU oracle (C port of cDBSCAN2) on this box's host cores, one process per chromosome
"""
import argparse
import ctypes
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
N_TOTAL = 200000000
MODE3 = ([5000, 7500, 10000], [50, 40, 30, 20])        # cLoops/pipe.py:337-340
CFG = 3                                                 # seed family of cloops_amd.synth.synth_genome
VARIANT = "v2"
HBM_PEAK_GBS = 8000.0                                   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
N_5M, CHR1_LEN, EPS_5M, MINPTS_5M = 5000000, 248956422, 2000, 5


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-total", type=float, default=N_TOTAL, help="PETs of the synthetic genome (tests use less)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU ranks (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-cpu-baseline", action="store_true",
                    help="time the sequential oracle over the WHOLE sweep (12 runs x 23 chromosomes, one process per chromosome: about a minute on "
                         "23+ cores) instead of the bounded 2-run sample, and check its chain against the GPU's")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-with-labels", action="store_true", help="skip the label-inclusive sweep (labels + tables on the host every run)")
    ap.add_argument("--proxy-ranks", default="2,4,8", help="single-GPU scaling proxy: time every rank's LPT share of N alone, for each N of the list (0 = off)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


PREFLIGHT_CHILD = r"""
import sys
import numpy as np
rank, world, local, tag = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
from cloops_amd.comm import Comm
c = Comm(rank, world, local, tag=tag)
v = c.allsum(np.arange(6, dtype=np.float64) + 1.0)
assert abs(float(v[0]) - world) < 1e-9 and abs(float(v[5]) - 6.0 * world) < 1e-9, v
k = c.allsum(np.asarray([rank + 1], dtype=np.int64))
assert int(k[0]) == world * (world + 1) // 2, k
tabs = c.gather_tables(np.full((rank + 1, 4), rank, np.int32), dst=0)
if rank == 0:
    assert [len(t) for t in tabs] == [r + 1 for r in range(world)] and all(int(t[0, 0]) == r for r, t in enumerate(tabs)), tabs
c.barrier()
c.close()
"""


def rccl_preflight(rank, world, local_rank, timeout=None, child=None):
    """The RCCL-direct exchanges of this run (libcloops_comm.so: communicator from the id file, all-reduce of the step vector,
    gather of variable-length tables, barrier) tried once in a CHILD process per rank before the run commits to them -- the
    parent has loaded neither HIP runtime yet, so the torch.distributed path is still open to it.  Every rank leaves its
    child's outcome in a file named after the launch; a rank reads all `world` of them: the decision is the same everywhere
    (all children fine -> RCCL direct, otherwise torch.distributed), so the ranks cannot split between two comm stacks.
    `child`: another script in the child's place (tests).  -> (ok, note)"""
    from cloops_amd import comm as cm                    # (importing the module loads no library)
    timeout = float(os.environ.get("CLOOPS_BENCH_PREFLIGHT_TIMEOUT", "120")) if timeout is None else timeout
    tag = cm.default_tag() + "_pf"
    mark = os.path.join(cm.ID_DIR, "cloops_comm_pf_%s" % tag)
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    note, ok = "", False
    for stale in ("%s.%d" % (mark, rank),) + ((cm.id_path(tag),) if rank == 0 else ()):
        try:
            os.remove(stale)                               # what an earlier launch under the same tag may have left behind
        except OSError:
            pass
    oldest = cm._process_start_time() - cm.STALE_ID_SKEW_S
    try:
        out = subprocess.run([sys.executable, "-c", child or PREFLIGHT_CHILD, str(rank), str(world), str(local_rank), tag], env=env, cwd=ROOT,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
        ok = out.returncode == 0
        if not ok:
            note = "rank %d: exit %d: %s" % (rank, out.returncode, out.stdout.decode("utf-8", "replace")[-300:].replace("\n", " | "))
    except subprocess.TimeoutExpired:
        note = "rank %d: no answer within %.0f s" % (rank, timeout)
    tmp = "%s.%d.tmp" % (mark, rank)
    with open(tmp, "w") as fh:
        fh.write("1" if ok else "0 " + note)
    os.replace(tmp, "%s.%d" % (mark, rank))
    # every rank's outcome (a rank whose child hung has waited `timeout` itself: the others wait that long and a little more)
    verdicts, t0 = {}, time.time()
    while len(verdicts) < world and time.time() - t0 < timeout + 60.0:
        for r in range(world):
            if r not in verdicts:
                try:
                    with open("%s.%d" % (mark, r)) as fh:
                        txt = fh.read()
                        if os.fstat(fh.fileno()).st_mtime >= oldest:      # (not a verdict of an earlier launch)
                            verdicts[r] = txt
                except (IOError, OSError):
                    pass
        if len(verdicts) < world:
            time.sleep(0.02)
    import atexit
    atexit.register(lambda path="%s.%d" % (mark, rank): os.path.exists(path) and os.remove(path))      # (every rank has read it long before)
    if rank == 0:
        atexit.register(lambda path=cm.id_path(tag): os.path.exists(path) and os.remove(path))           # the child's id file, if it was killed before its unlink
    all_ok = len(verdicts) == world and all(v.startswith("1") for v in verdicts.values())
    bad = "; ".join(v[2:] for v in verdicts.values() if not v.startswith("1")) or ("missing verdicts of ranks %s" % sorted(set(range(world)) - set(verdicts)) if len(verdicts) < world else "")
    return all_ok, bad


def main(argv=None):
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not launched by torch.distributed.run: start the N ranks ourselves (one process per GPU)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)]
        cmd += sys.argv[1:] if argv is None else list(argv)
        sys.exit(subprocess.call(cmd))

    # stdout carries exactly one JSON line (rank 0): park the real stdout and point fd 1 at stderr until the end
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("CLOOPS_BENCH_FORCE_DIST") == "1"
    on_gpu = args.backend == "nccl"
    if os.environ.get("CLOOPS_BENCH_PRELOAD"):
        # test hook: a module imported before anything else (the CPU tests install their stand-in GPU backend here)
        importlib.import_module(os.environ["CLOOPS_BENCH_PRELOAD"])
    import numpy as np
    torch = dist = tdev = comm = None
    # GPU ranks talk over RCCL through libcloops_comm.so -- no PyTorch in the process (torch.distributed.run is only the
    # launcher): with torch imported its bundled HIP runtime serves libcloops_hip.so as well and a sweep runs ~6 % slower.
    # CLOOPS_BENCH_COMM=torch keeps the torch.distributed path; the CPU tests (--backend gloo) always use it.
    direct = use_dist and on_gpu and os.environ.get("CLOOPS_BENCH_COMM", "rccl") != "torch"
    comm_note = None
    comm_so = os.path.join(ROOT, "cloops_amd", "libcloops_comm.so")
    if direct and not os.path.exists(comm_so):
        # the ONLY fall-back: the library was never built.  Every rank sees the same file system, so every rank takes the same
        # branch -- and nothing of either runtime has been loaded yet.  Once libcloops_comm.so is mapped a failure is fatal on
        # purpose: a rank that quietly moved to torch.distributed would leave the others waiting in an RCCL collective.
        comm_note = "libcloops_comm.so not built, torch.distributed instead"
        sys.stderr.write("bench.py: %s\n" % comm_note)
        direct = False
    if direct and (world > 1 or os.environ.get("CLOOPS_BENCH_PREFLIGHT") == "force") and os.environ.get("CLOOPS_BENCH_PREFLIGHT", "1") != "0":
        # RCCL through libcloops_comm.so has to work on THIS node for every rank before the run depends on it (its failures --
        # a rank that cannot open its IPC handles, a communicator that never forms -- are hangs, not errors): tried in child
        # processes, decided by all ranks alike, while this process can still take the torch.distributed path
        ok, why = rccl_preflight(rank, world, local_rank)
        if not ok:
            comm_note = "RCCL-direct pre-flight failed (%s): torch.distributed instead" % why
            if rank == 0:
                sys.stderr.write("bench.py: %s\n" % comm_note)
            direct = False
    if direct:
        from cloops_amd.comm import Comm
        comm = Comm(rank, world, local_rank)
    elif use_dist:
        # torch first: its bundled HIP runtime then also serves libcloops_hip.so (same SONAME; the other order aborts
        # at the first RCCL call)
        import torch
        import torch.distributed as dist
        if on_gpu:
            torch.cuda.set_device(local_rank)
            tdev = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=tdev)
        else:
            dist.init_process_group("gloo")
    os.environ["CLOOPS_DEVICES"] = str(local_rank if (use_dist and on_gpu) else 0)
    from cloops_amd import api, pipe
    from cloops_amd.synth import synth_chrom, chrom_sizes
    from cloops_amd.dist import gather_tables, make_allsum, lpt_assign

    def sync_all():
        if comm is not None:
            comm.barrier()                              # hipDeviceSynchronize + a barrier over the ranks
        elif use_dist:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    # ---- the genome, LPT-sharded over the ranks -------------------------------------------------------------
    n_total = int(args.n_total)
    sizes = chrom_sizes(n_total)
    mine = sorted(lpt_assign([n for _, _, n in sizes], world)[rank])
    device = local_rank if (use_dist and on_gpu) else 0
    t_gen = time.perf_counter()
    fs = []
    for ci in mine:
        name, length, n = sizes[ci]
        X, Y = synth_chrom(n, length, 1000 * CFG + ci)
        fs.append(pipe.CACHE.put_arrays("%s-%s" % (name, name), X, Y, device=device))
    t_gen = time.perf_counter() - t_gen
    # the per-run exchange of the chained cut is ONE vector of ~48 KB of statistics per step: over RCCL through a pinned
    # staging buffer (a ring over xGMI, tens of microseconds; CLOOPS_BENCH_STATS=gloo keeps it on the host over a gloo group:
    # TCP loopback, hundreds of microseconds per ring at 8 ranks); RCCL also carries the final gather of the candidate tables
    stats_gloo = os.environ.get("CLOOPS_BENCH_STATS", "rccl") == "gloo"
    stats_group = dist.new_group(backend="gloo") if (use_dist and comm is None and on_gpu and stats_gloo) else None
    if not use_dist:
        allsum = None
    elif comm is not None:
        allsum = comm.make_allsum()
    elif on_gpu and not stats_gloo:
        allsum = make_allsum(device=tdev)
    else:
        allsum = make_allsum(device=None, group=stats_group)
    eps_list, minpts_list = MODE3

    # K2 is timed inside a sweep on the largest local chromosome (chr1 on rank 0): HIP events on the library's stream -- in the
    # last WARM-UP sweep only: the event records between the kernels of that stream cost the sweep ~7 % (0.214 -> 0.230 s), so
    # the timed sweeps run without them (the roofline figure comes from the solo replay behind the timed region anyway)
    probe_f = max(fs, key=lambda f: len(pipe.CACHE.get(f))) if fs else None
    k2_log = []

    def probe(f, ep, m, cut_in, res):
        if f == probe_f and res.timing is not None:
            k2_log.append((ep, m, cut_in, dict(res.timing)))

    def one_sweep(log_k2):
        # the path's final exchange (cLoops/pipe.py:119-127 merges its workers' results): the candidate tables go to rank 0 from
        # where cl_cand_finish_device left them -- exact sizes over RCCL, one device-to-host copy at the root, none elsewhere; run as
        # the sweep's device_consumer, i.e. while the handles are still pinned and locked
        gather = (lambda d: comm.gather_device([v["dev_rows"] for v in d.values()], [v["n_rows"] for v in d.values()], dst=0, copy=False)) if comm is not None else None
        dataI, cut, cuts, steps = pipe.runSweepFast(fs, eps_list, minpts_list, cut=0, variant=VARIANT, allsum=allsum,
                                                    probe=probe if (log_k2 and rank == 0) else None, finish_device=comm is not None,
                                                    device_consumer=gather)
        if comm is not None:
            ncand = sum(len(t) for t in dataI.gathered)
        elif use_dist:
            rows = [v["boxes"] for v in dataI.values() if len(v["boxes"])]
            rows = rows if rows else np.zeros((0, 4), np.int32)
            tabs = comm.gather_tables(rows, dst=0, copy=False) if comm is not None else gather_tables(rows, device=tdev, dst=0, copy=False)
            ncand = sum(len(t) for t in tabs)
        else:
            ncand = sum(len(v["boxes"]) for v in dataI.values())
        return cut, steps, ncand

    first_sweep_s = None
    for w in range(args.warmup):
        probing = w == args.warmup - 1 and probe_f is not None and rank == 0
        if probing:
            pipe.CACHE.get(probe_f).chrom.set_profiling(True)
        t0 = time.perf_counter()
        one_sweep(probing)
        if w == 0:
            first_sweep_s = time.perf_counter() - t0
        if probing:
            pipe.CACHE.get(probe_f).chrom.set_profiling(False)
    sync_all()
    t0 = time.perf_counter()
    pets = 0
    for k in range(args.steps):
        cut, steps, ncand = one_sweep(False)
        pets += sum(s["n_in"] for s in steps)          # genome-wide (all-reduced inside runSweepFast)
    sync_all()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.allmax(elapsed)
    elif use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev if on_gpu else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    line = None
    if rank == 0:
        line = {
            "metric": "PETs clustered/sec (whole node) + wall-clock for mode-3 eps x minPts sweep",
            "value": pets / elapsed,
            "unit": "PETs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "sweep_wall_s": elapsed / max(1, args.steps),
            "first_sweep_s": first_sweep_s,
            "comm": ("rccl %d (libcloops_comm.so built against %d, no torch in the process)" % (comm.rccl_loaded, comm.rccl_built) if comm is not None else
                     (("torch.distributed/%s" % args.backend) + ("; " + comm_note if comm_note else "") if use_dist else None)),
            "config": {"workload": "synthetic-%s-23chr-mode3 (BASELINE.json configs[3])" % _human(n_total),
                       "variant": "cDBSCAN2", "pets": n_total, "chromosomes": len(sizes), "eps": eps_list, "minPts": minpts_list,
                       "runs_per_sweep": len(steps), "chained_cut": True, "cuts": [s.get("cut_out") for s in steps],
                       "final_cut": int(cut), "candidate_loops": int(ncand),
                       "pets_entering_dbscan_per_sweep": int(pets // max(1, args.steps)),
                       "parallelism": "23 chromosomes LPT-sharded over %d GPU(s)" % world,
                       "synthesis_s_rank0": round(t_gen, 2)},
        }
        if probe_f is not None and on_gpu:
            # replay the sweep's 12 runs IN THE SWEEP'S ORDER on the probe chromosome alone (3 passes): the kernels without
            # neighbours on the device, with the region query re-used inside every eps exactly as a sweep re-uses it, then
            # the same passes with the re-use switched off (every run its own full region query)
            r = pipe.CACHE.get(probe_f)
            settings = [(st["eps"], st["minPts"], st["cut_in"]) for st in steps]
            line["roofline"] = roofline_block(k2_replay(r.chrom, settings, sorted(set(minpts_list)), 3), len(r))
            if k2_log:
                line["roofline"]["in_sweep_avg_launch_ms"] = sum(max(t[3]["ms_region"] - t[3]["ms_bracket"], 1e-6) for t in k2_log) / len(k2_log)
                line["roofline"]["in_sweep_source"] = "HIP events around K2 on chr1's stream during the last warm-up sweep (same work as a timed one; the event records between the kernels cost a sweep ~7 %, so the timed sweeps run without them)"
    # single-GPU legs behind the timed region: the one-dataset sweep, the label-inclusive form of the same sweep, the scaling proxy
    if rank == 0 and world == 1 and on_gpu:
        # a dataset's ONLY sweep: every handle forgets its q index, fine layout, last layout and cached counts (allocations stay)
        cold = []
        for _ in range(2):
            for f in fs:
                pipe.CACHE.get(f).chrom.drop_indexes()
            t0c = time.perf_counter()
            one_sweep(False)
            cold.append(time.perf_counter() - t0c)
        line["cold_sweep_s"] = min(cold)
        line["cold_sweep_note"] = ("the same sweep with cl_chrom_drop_indexes on every chromosome first (q index 4 radix passes + fine layout 2 passes + "
                                   "everything a timed sweep does; handles and workspaces allocated): what a dataset swept once pays; best of 2")
        if not args.no_with_labels:
            line["with_labels"] = with_labels_sweep(pipe, fs, steps, pets // max(1, args.steps))
            line["with_labels_sweep_s"] = line["with_labels"]["sweep_wall_s"]
        plist = [int(x) for x in str(args.proxy_ranks).split(",") if x.strip() and int(x) > 1]
        if plist:
            # the 1 / 2 / 4 / 8 table north_star asks for, as far as ONE GPU can evidence it: the largest N in full, the others compact
            prox = {n: scaling_proxy(pipe, lpt_assign, fs, sizes, steps, n, elapsed / max(1, args.steps)) for n in sorted(plist)}
            top = max(prox)
            line["scaling_proxy"] = prox[top]
            line["scaling_proxy"]["by_ranks"] = {str(n): {k: v.get(k) for k in ("makespan_s", "sum_over_steps_of_slowest_rank_s", "predicted_sweep_s", "predicted_speedup", "lpt_balance")} for n, v in prox.items()}
    # the secondary single-eps figure and the CPU baseline: rank 0, single GPU only
    if rank == 0 and world == 1 and on_gpu:
        pipe.CACHE.clear()
        if not args.no_secondary:
            line["secondary_5M"] = secondary_5m(api, synth_chrom)
            line["secondary_chr21_cli"] = chr21_cli()
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_full(n_total, steps, int(cut), int(ncand)) if args.full_cpu_baseline else cpu_baseline(api, sizes, steps)
    pipe.CACHE.clear()
    if comm is not None:
        comm.barrier()
        comm.close()
    elif use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # everything written to fd 1 during the run (RCCL's banner comes through C stdio) went to stderr: give stdout
    # back and print the ONE JSON line
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(line), flush=True)


def _human(n):
    return "%dM" % (n // 1000000) if n >= 1000000 else "%dk" % (n // 1000)


def k2_bytes(tm):
    """sorted (q, p) in + count out per PET that entered DBSCAN + the strip offset table"""
    return int(tm["n_in"]) * 12 + int(tm["n_strips"]) * 4


def k2_replay(chrom, settings, served, passes=3):
    """The sweep's runs, in the sweep's order, on ONE resident chromosome alone on the GPU, `passes` times, HIP events
    around the kernels (the library's own brackets): first with the region query re-used inside an eps as the sweep
    driver runs it (count cache, the sweep's minPts list announced: `served`), then with the re-use off.  -> {"reuse": rows, "full": rows}, a row =
    (eps, minPts, cut, timing dict, region mode 0 / 1 / 2 of cl_last_region_mode)."""
    out = {}
    chrom.set_profiling(True)
    only = os.environ.get("CLOOPS_REPLAY_ONLY")                          # "reuse" / "full": one kind of pass (tools/profile_bench.sh)
    for key, on in [kv for kv in (("reuse", True), ("full", False)) if only in (None, "", kv[0])]:
        chrom.set_count_reuse(on)
        chrom.set_count_thresholds(served if on else [])
        chrom.set_eps_list(sorted(set(ep for ep, _, _ in settings)) if on else [])     # (as the sweep driver announces it)
        rows = []
        for p in range(passes + 1):                     # (pass 0 warms the handle up: allocations, the q index)
            for ep, m, cut in settings:
                chrom.cluster_async(VARIANT, ep, m, cut, want_labels=False, want_boxes=False)
                mode = chrom.last_region_mode()
                res = chrom.wait()
                if p > 0:
                    rows.append((ep, m, cut, dict(res.timing), mode))
        out[key] = rows
    chrom.set_count_reuse(True)
    chrom.set_count_thresholds(served)
    chrom.set_eps_list([])
    chrom.set_profiling(False)
    if only:
        out["full" if only == "reuse" else "reuse"] = out[only]
    return out


def roofline_block(replay, n_probe):
    """The region query (K2) on the probe chromosome, three ways:
      frac / achieved   AMORTISED over the runs of an eps the way the sweep executes it: the first run of an eps queries the whole
                        base layout (k_region_keys, counts bracketed for the sweep's minPts list), every run under a cut queries its
                        cut band only (k_band, a kernel of its own between its own pair of events) and reads every other PET's word
                        in place: sum over the 12 runs of SURVEY 8d's algorithmic bytes N * 12 + (S + 2) * 4 / sum over the 12 runs
                        of (k_region_keys time + k_band time).  (Traversal levels <= 3 carry the words through the cut compaction
                        instead: that cost = the difference of the sort-phase brackets against the pass without re-use.)
      per_launch        the k_region_keys launches that executed, each against the algorithmic bytes of the PETs it covered
      full_query        every run its own full query (cl_set_count_reuse(0)): the per-launch figure of rounds 1-3
    K2 is the only kernel between its two events; the bracket around an EMPTY kernel (event packets + dispatch gap, calibrated by
    the library) is taken out of every launch -- rocprofv3's kernel duration has no such term (profiles/README.md)."""
    rows, full = replay["reuse"], replay["full"]

    def net(tm):
        return max(tm["ms_region"] - tm["ms_bracket"], 0.0)

    def band(tm):
        # traversal level 4: the band query of a re-using run is a kernel of its own (k_band), bracketed by its own pair of events
        return max(tm.get("ms_band", 0.0) - tm["ms_bracket"], 0.0) if tm.get("ms_band", 0.0) > 0 else 0.0

    def agg(rs):
        return (sum(k2_bytes(r[3]) for r in rs), sum(r[3]["ms_region"] for r in rs),
                sum(max(net(r[3]) if r[3].get("n_queried", 1) else 0.0, 0.0) + band(r[3]) for r in rs))
    # levels <= 3: the carry cost of a re-using run = its sort-phase bracket (cut compaction + words) minus the plain compaction's of
    # the same run; level 4 copies nothing and brackets its band query itself (ms_band): no carry
    carry = [max(a[3]["ms_sort"] - f[3]["ms_sort"], 0.0) if (a[4] == 2 and not a[3].get("ms_band", 0.0) > 0) else 0.0 for a, f in zip(rows, full)]
    b, raw, k2net = agg(rows)
    tot = max(k2net + sum(carry), 1e-9)
    ach = b / (tot * 1e-3) / 1e9
    fb, _, fnet = agg(full)
    # the launches of k_region_keys that actually ran (the hardware figure next to the amortised one): bytes of the layout each covered
    launched = [r for r in rows if r[3].get("n_queried", 0) > 0 and net(r[3]) > 0]
    lb = sum(int(r[3]["n_queried"]) * 12 + int(r[3]["n_strips"]) * 4 for r in launched)
    lt = sum(net(r[3]) for r in launched)
    per_launch = {"launches": len(launched), "bytes": lb // max(1, len(launched)), "avg_launch_ms": lt / max(1, len(launched)),
                  "achieved": lb / max(lt * 1e-3, 1e-12) / 1e9, "frac": lb / max(lt * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                  "note": "the k_region_keys launches that executed (one per eps: the whole base layout, counts bracketed for the sweep's minPts list), "
                          "algorithmic bytes of the PETs each covered / its own duration"}
    per_eps = {}
    for ep in sorted({r[0] for r in rows}):
        sel = [k for k, r in enumerate(rows) if r[0] == ep]
        bb, _, nn = agg([rows[k] for k in sel])
        cc = sum(carry[k] for k in sel)
        _, _, ff = agg([full[k] for k in sel])
        make = [net(rows[k][3]) for k in sel if rows[k][3].get("n_queried", 0) > 0]
        bnd = [band(rows[k][3]) if rows[k][3].get("ms_band", 0.0) > 0 else net(rows[k][3]) for k in sel if rows[k][4] == 2 or rows[k][3].get("ms_band", 0.0) > 0]
        per_eps[str(ep)] = {"achieved": bb / ((nn + cc) * 1e-3) / 1e9, "frac": bb / ((nn + cc) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "first_run_launch_ms": sum(make) / max(1, len(make)), "band_launch_ms": sum(bnd) / max(1, len(bnd)),
                            "carry_ms": cc / max(1, len(bnd)), "full_query_avg_launch_ms": ff / len(sel)}
    traffic, src = None, None
    tpath = os.path.join(ROOT, "profiles", "k2_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as fh:
                tj = json.load(fh)
            traffic, src = tj.get("hbm_bytes_per_launch"), "profiles/k2_traffic.json (%s; %s)" % (tj.get("workload"), tj.get("source"))
        except Exception:
            pass
    n_make = len(launched)
    n_band = len([r for r in rows if r[4] == 2 or r[3].get("ms_band", 0.0) > 0])
    band_ms = [band(r[3]) if r[3].get("ms_band", 0.0) > 0 else (net(r[3]) if r[4] == 2 else None) for r in rows]
    band_ms = [x for x in band_ms if x is not None]
    return {"bound": "hbm", "kernel": "k_region_keys (the sorted-key region query: once per eps, on the base layout) + k_band (the cut band of every run under a cut)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
            "amortised": True, "launches": len(rows), "runs_with_full_query": n_make, "runs_on_the_band": n_band,
            "probe": "chr1 of the genome (%d PETs) alone on the GPU, the sweep's 12 (eps, minPts, cut) runs in the sweep's order x 3 passes" % n_probe,
            "algorithmic_bytes_per_launch": b // len(rows), "avg_launch_ms": tot / len(rows),
            "first_run_avg_launch_ms": lt / max(1, n_make),
            "band_avg_launch_ms": sum(band_ms) / max(1, len(band_ms)),
            "carry_avg_ms": sum(carry) / max(1, n_band),
            "avg_event_bracket_ms": raw / len(rows), "empty_kernel_bracket_ms": float(rows[0][3]["ms_bracket"]),
            "per_launch": per_launch,
            "full_query": {"achieved": fb / (fnet * 1e-3) / 1e9, "frac": fb / (fnet * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": fnet / len(full),
                           "note": "every run its own full region query (cl_set_count_reuse(0)): the per-launch figure of rounds 1-3"},
            "per_eps": per_eps}


def with_labels_sweep(pipe, fs, steps, pets_per_sweep, reps=2):
    """SURVEY.md 8d(1)'s end point on the headline workload: the same 12 runs (the chain's own eps, minPts, cut) over the same
    23 resident chromosomes, every run landing its labels and its cluster table in pinned host memory (what
    cLoops/pipe.py:70-102 consumes) -- which the sweep's own form (statistics and candidates on the device) never pays.
    Two forms: the labels as the reference holds them, CLUSTERED points only (`.labels` is a dict of them, cDBSCAN2.py:186-191): one
    (row, label) pair per labelled PET, written by the label kernel straight into page-locked memory (cl_cluster_pairs_async) -- the
    figure; and row-aligned int32 labels of every PET (`row_aligned`: the row-order scatter and 0.8 GB over PCIe per run).
    All chromosomes of a run are enqueued, then collected."""
    res = [pipe.CACHE.get(f) for f in fs]
    res.sort(key=lambda r: -len(r))

    # the cuts are given (the chain of the timed sweeps), so consecutive runs are independent: a chromosome has two result slots, and
    # run k + 1 of every chromosome is enqueued before run k is collected -- its kernels execute while run k's labels cross PCIe
    def sweep_pairs():
        nlab = 0
        for k, st in enumerate(steps):
            for r in res:
                r.chrom.cluster_pairs_async(VARIANT, st["eps"], st["minPts"], st["cut_in"], want_boxes=True)
            if k > 0:
                for r in res:
                    nlab += int(r.chrom.wait_pairs(defer=True)[1].shape[0])      # (their copies run side by side ...)
                for r in res:
                    r.chrom.pairs_sync()                                          # (... and are all on the host here)
        for r in res:
            nlab += int(r.chrom.wait_pairs(defer=True)[1].shape[0])
        for r in res:
            r.chrom.pairs_sync()
        return nlab

    def sweep_mask():
        nlab = 0
        for k, st in enumerate(steps):
            for r in res:
                r.chrom.cluster_rowmask_async(VARIANT, st["eps"], st["minPts"], st["cut_in"], want_boxes=True)
            if k > 0:
                for r in res:
                    nlab += int(r.chrom.wait_rowmask(defer=True)[2].shape[0])
                for r in res:
                    r.chrom.pairs_sync()
        for r in res:
            nlab += int(r.chrom.wait_rowmask(defer=True)[2].shape[0])
        for r in res:
            r.chrom.pairs_sync()
        return nlab

    def sweep_rows():
        nlab = 0
        for k, st in enumerate(steps):
            for r in res:
                r.chrom.cluster_async(VARIANT, st["eps"], st["minPts"], st["cut_in"], want_labels=True, want_boxes=True)
            if k > 0:
                for r in res:
                    nlab += int(r.chrom.wait().labels.shape[0])
        for r in res:
            nlab += int(r.chrom.wait().labels.shape[0])
        return nlab

    def timed(fn):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            k = fn()
        return (time.perf_counter() - t0) / reps, k
    dt_pairs, nlab_pairs = timed(sweep_pairs)
    dt, nlab = timed(sweep_mask)
    assert nlab == nlab_pairs, (nlab, nlab_pairs)
    mask_bytes = len(steps) * sum(8 * ((len(r) + 63) // 64) for r in res)
    for r in res:
        r.chrom.set_device_labels(True)
    dt_rows, nrows = timed(sweep_rows)
    for r in res:
        r.chrom.set_device_labels(False)
    return {"sweep_wall_s": dt, "value": pets_per_sweep / dt, "unit": "PETs/s", "runs": len(steps), "labelled_pets_to_host_per_sweep": nlab,
            "bytes_to_host_per_sweep": 4 * nlab + mask_bytes,
            "end_point": "the CLUSTERED PETs of every run (the reference's `.labels` holds nothing else) as one bit per input row + their int32 labels in "
                         "ascending row order (cl_cluster_rowmask_async) + the cluster table of every run in pinned host memory (SURVEY.md 8d(1)); "
                         "cuts = the chain of the timed sweeps; no statistics / candidate work on the device",
            "pairs": {"sweep_wall_s": dt_pairs, "value": pets_per_sweep / dt_pairs, "bytes_to_host_per_sweep": 8 * nlab_pairs,
                      "end_point": "the same set as (row, label) int32 pairs in no particular order (cl_cluster_pairs_async): 8 bytes per clustered PET, PCIe-bound"},
            "row_aligned": {"sweep_wall_s": dt_rows, "value": pets_per_sweep / dt_rows, "labels_to_host_per_sweep": nrows, "bytes_to_host_per_sweep": 4 * nrows,
                            "end_point": "row-aligned int32 labels of EVERY PET (-1 = not clustered) + the cluster table of every run"}}


_PROBE = None


def _d2h_rate_gbs():
    """GB/s of one 4.8 MB device-to-host copy into page-locked memory right now (libamdhip64 through ctypes; buffers kept)"""
    global _PROBE
    nb = 4800000
    if _PROBE is None:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        h, d = ctypes.c_void_p(), ctypes.c_void_p()
        if hip.hipHostMalloc(ctypes.byref(h), nb, 0) != 0 or hip.hipMalloc(ctypes.byref(d), nb) != 0:
            raise RuntimeError("copy probe: allocation failed")
        _PROBE = (hip, h, d)
    hip, h, d = _PROBE
    hip.hipDeviceSynchronize()
    hip.hipMemcpy(h, d, nb, 2)
    t0 = time.perf_counter()
    hip.hipMemcpy(h, d, nb, 2)
    return nb / (time.perf_counter() - t0) / 1e9


def settle_copies(limit_s=3.0, good_gbs=35.0):
    """Waits until PCIe copies run at full rate again -> seconds waited.  Measured on MI355X / ROCm 7.2 (tools/d2h_stream_probe.cpp,
    tools/gather_probe.py and this probe along a proxy run): for 0.41 s after a handle's buffers are freed EVERY host <-> device
    copy of the process, on any stream, into any page-locked buffer, runs at 14 GB/s instead of 50.  The proxy frees the handles
    of one rank's share before it makes the next (a real rank never does), so without this wait the gather measured right behind
    the last share took 4.2 ms instead of 1.3."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < limit_s:
        if _d2h_rate_gbs() >= good_gbs and _d2h_rate_gbs() >= good_gbs:
            break
        time.sleep(0.02)
    return time.perf_counter() - t0


def scaling_proxy(pipe, lpt_assign, fs, sizes, steps, nranks, one_gpu_sweep_s, reps=3):
    """Single-GPU evidence for the N-GPU claim.  Every rank's LPT share of `nranks` is swept ALONE on this GPU, replaying the
    genome-wide chain (the cuts a real run all-reduces), and timed step by step.  A real run meets after every step (the cut
    is a genome-wide estimate: one all-reduce per run), so what bounds it is the SUM over the steps of the slowest rank's
    step, not the slowest rank's sum; on top come the exchanges themselves, timed here through libcloops_comm.so at world
    size 1 (staging through pinned memory + the RCCL call + one stream synchronisation; no xGMI hop): 13 all-reduces of the
    step vector and the gather of the candidate tables at the root.  `predicted_sweep_s` is the sum of those terms."""
    import numpy as np
    shares = lpt_assign([n for _, _, n in sizes], nranks)
    forced = [s.get("cut_out") for s in steps]
    eps = sorted({s["eps"] for s in steps})
    mps = sorted({s["minPts"] for s in steps}, reverse=True)
    per, walls, tails, tables = [], [], [], []
    for r, share in enumerate(shares):
        fr = [fs[ci] for ci in sorted(share)]
        if not fr:
            per.append({"rank": r, "chromosomes": [], "pets": 0, "sweep_wall_s": 0.0})
            walls.append([0.0] * len(steps)); tails.append(0.0)
            continue
        # the share as a rank would hold it: handles of its own, made largest first (a rank's stream pool spreads ITS chromosomes
        # over the shared streams; the handles of the whole genome sit on the streams the 23-chromosome load balancing gave them,
        # which can put a share's chromosomes on one stream)
        copies = {}
        for f in sorted(fr, key=lambda f: -len(pipe.CACHE.get(f))):
            src = pipe.CACHE.get(f)
            name = "mem://rank_share/%s-%s" % (src.key[0], src.key[1])
            pipe.CACHE.put_chrom(name, pipe._make_chrom(src.X, src.Y, src.device), src.X, src.Y, ids=src.ids, key=src.key, device=src.device)
            copies[f] = name
        fr = [copies[f] for f in fr]
        try:
            pipe.runSweepFast(fr, eps, mps, cut=0, variant=VARIANT, forced_cuts=forced)
            # the FASTEST of `reps` sweeps: a share is 14-19 ms of wall clock, and one hiccup of the host (a page fault, a thread
            # that wakes late) in one of two sweeps used to move the slowest rank -- the figure the prediction is built on -- by 20 %
            acc, tot = None, None
            for _ in range(reps):
                t0 = time.perf_counter()
                res = pipe.runSweepFast(fr, eps, mps, cut=0, variant=VARIANT, forced_cuts=forced)
                t = time.perf_counter() - t0
                if tot is None or t < tot:
                    tot, acc = t, np.asarray([st["wall_s"] for st in res[3]])
            tables += [v["boxes"] for v in res[0].values() if len(v["boxes"])]
        finally:
            for name in fr:
                pipe.CACHE.drop(name)
        dt = tot
        walls.append(list(acc)); tails.append(max(dt - float(acc.sum()), 0.0))                     # tail: candidate dedup + tables to the host
        per.append({"rank": r, "chromosomes": [sizes[ci][0] for ci in sorted(share)], "pets": int(sum(sizes[ci][2] for ci in share)),
                    "sweep_wall_s": dt, "steps_wall_s": [round(x, 6) for x in walls[-1]]})
    mk = max(p["sweep_wall_s"] for p in per)
    sum_of_max = float(np.max(np.asarray(walls), axis=0).sum()) + max(tails)
    out = {"ranks": nranks, "per_rank": per, "makespan_s": mk, "sum_over_steps_of_slowest_rank_s": sum_of_max, "one_gpu_sweep_s": one_gpu_sweep_s,
           "implied_speedup": one_gpu_sweep_s / mk if mk > 0 else None, "implied_efficiency": one_gpu_sweep_s / mk / nranks if mk > 0 else None,
           "lpt_balance": max(p["pets"] for p in per) / (sum(p["pets"] for p in per) / float(nranks)),
           "note": "each rank's share timed ALONE on one MI355X with the genome-wide cut chain forced (runSweepFast(forced_cuts)), the fastest of 3 sweeps; makespan_s = the slowest "
                   "rank's whole sweep (no meeting between the steps: a lower bound); sum_over_steps_of_slowest_rank_s = with the per-step meeting a real run has; "
                   "every share runs on handles of its own (made for the proxy, as a rank would make them), not on the handles of the 23-chromosome sweep"}
    try:
        from cloops_amd.comm import Comm
        waited = settle_copies()
        c1 = Comm(0, 1, 0)
        vec = np.zeros(8 + 3840 + 2048 + 6)               # the per-step statistics vector of runSweepFast(allsum=...)
        c1.allsum(vec)
        t0 = time.perf_counter()
        for _ in range(len(steps) + 1):
            c1.allsum(vec)
        t_ar = time.perf_counter() - t0
        # the gather: the candidate tables of the whole genome where a sweep's dedup leaves them (device memory of the 23 resident
        # handles) -> one page-locked buffer at the root (cl_comm_gather_device; at world size 1 every table is a direct
        # device-to-host copy, a real root receives the other ranks' rows over xGMI into a device buffer first)
        dres = pipe.runSweepFast(fs, eps, mps, cut=0, variant=VARIANT, forced_cuts=forced, finish_device=True)[0]
        ptrs, nrows = [v["dev_rows"] for v in dres.values()], [v["n_rows"] for v in dres.values()]
        c1.gather_device(ptrs, nrows, dst=0, copy=False)
        # the fastest of three, like the shares' sweeps -- and again (up to 8 rounds, a quarter of a second apart) while the copies
        # run far below the rate the probe sees: the slow-copy state after a release (settle_copies) has been seen to come back
        # for the communicator's stream alone
        t_gs, rounds = [], 0
        nbytes = 16 * int(sum(nrows))
        while True:
            rounds += 1
            for _ in range(3):
                t0 = time.perf_counter()
                ncand = sum(len(t) for t in c1.gather_device(ptrs, nrows, dst=0, copy=False))
                t_gs.append(time.perf_counter() - t0)
            if rounds >= 8 or min(t_gs) <= 2.0 * nbytes / (_d2h_rate_gbs() * 1e9) + 3e-4:
                break
            time.sleep(0.25)
            waited += settle_copies()
        t_g = min(t_gs)
        c1.close()
        out["exchanges_world1"] = {"allreduce_calls": len(steps) + 1, "allreduce_total_s": t_ar, "gather_rows": int(ncand), "gather_s": t_g, "gather_s_all": [round(x, 6) for x in t_gs],
                                   "d2h_gbs_at_measurement": round(_d2h_rate_gbs(), 1), "waited_for_full_copy_rate_s": round(waited, 3), "gather_rounds": rounds,
                                   "rccl": c1.rccl_loaded,
                                   "note": "through libcloops_comm.so on ONE rank: the all-reduce with host staging + RCCL call + stream synchronisation, the gather device-resident (cl_comm_gather_device); without the xGMI hops of a real ring"}
        out["predicted_sweep_s"] = sum_of_max + t_ar + t_g
        out["predicted_speedup"] = one_gpu_sweep_s / out["predicted_sweep_s"]
    except Exception as e:                                   # (no RCCL on this box: the compute terms stand alone)
        out["exchanges_world1"] = None
        out["exchanges_note"] = "libcloops_comm.so / RCCL not usable here: %s: %s" % (type(e).__name__, e)
        out["predicted_sweep_s"] = None
    return out


def secondary_5m(api, synth_chrom, steps=20, warmup=3):
    """BASELINE.json configs[1]: 5 M PETs on one chromosome, eps 2000, minPts 5 -- one clustering run per step,
    labels + table to the host, pipelined over the two result slots (the round-1 headline, kept comparable)."""
    import numpy as np
    X, Y = synth_chrom(N_5M, CHR1_LEN, 2000)
    ch = api.Chromosome(X, Y, device=0)
    ch.set_profiling(True)
    ch.set_layout_reuse(False)             # one (eps, minPts) repeated: every step must pay its whole run, sort included

    def run(nsteps, k2):
        res = None
        ch.cluster_async(VARIANT, EPS_5M, MINPTS_5M, 0)
        for k in range(nsteps):
            if k + 1 < nsteps:
                ch.cluster_async(VARIANT, EPS_5M, MINPTS_5M, 0)
            res = ch.wait()
            if k2 is not None:
                k2.append(dict(res.timing))
        return res
    run(warmup, None)
    t0 = time.perf_counter()
    k2 = []
    res = run(steps, k2)
    dt = time.perf_counter() - t0
    n_in = int(k2[-1]["n_in"])
    net = float(np.mean([max(t["ms_region"] - t["ms_bracket"], 1e-6) for t in k2]))
    b = k2_bytes(k2[-1])
    out = {"workload": "synthetic-5M-chr1-eps2000-minPts5 (BASELINE.json configs[1])", "value": n_in * steps / dt, "unit": "PETs/s",
           "ms_per_step": dt / steps * 1e3, "clusters": int(res.n_clusters),
           "k2_avg_launch_ms": net, "k2_achieved_GBs": b / (net * 1e-3) / 1e9, "k2_frac": b / (net * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "k2_algorithmic_bytes": b,
           "kernel_ms": {k[3:]: round(float(v), 4) for k, v in k2[-1].items() if k.startswith("ms_") and k != "ms_bracket"}}
    ch.close()
    return out


def chr21_cli():
    """BASELINE.json configs[0] as the reference runs it (examples/run.sh:1: `cLoops -f <chr21 BEDPE> -o out -m 1`): wall time of
    `python -m cloops_amd -f <bedpe.gz> -o out -m 1` as a child process -- interpreter start, BEDPE parse, upload, the chained
    sweep (eps 500 / 1000 / 2000, minPts 5), significance test, `.loop` file -- beside BASELINE.md section 2's 9.8 s for the
    reference (one core, converted copy).  The BEDPE is written from the committed golden mid-points of the example
    (tests/golden/chr21_input.npz: the same 99 674 PETs); the `.loop` file is compared with the golden table."""
    import gzip
    import shutil
    import tempfile
    gold = os.path.join(ROOT, "tests", "golden")
    try:
        import numpy as np
        z = np.load(os.path.join(gold, "chr21_input.npz"))
        X, Y = z["X"], z["Y"]
    except Exception as e:
        return {"skipped": "no golden input: %s" % e}
    d = tempfile.mkdtemp(prefix="cloops_cli_")
    try:
        bed = os.path.join(d, "chr21.bedpe.gz")
        with gzip.open(bed, "wt") as fh:
            for x, y in zip(X.tolist(), Y.tolist()):
                fh.write("chr21\t%d\t%d\tchr21\t%d\t%d\tid\t1\t+\t-\n" % (x, x, y, y))
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        walls = []
        for rep in range(2):                              # (the second run has the page cache and the GPU context warm)
            out = os.path.join(d, "run%d" % rep)
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, "-m", "cloops_amd", "-f", bed, "-o", out, "-m", "1"], env=env, cwd=d,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            walls.append(time.perf_counter() - t0)
            if p.returncode != 0:
                return {"error": p.stdout.decode("utf-8", "replace")[-400:]}
        got = open(os.path.join(d, "run1.loop")).read()
        want_path = os.path.join(gold, "chr21_v2.loop")
        same = os.path.exists(want_path) and got == open(want_path).read()
        return {"workload": "chr21 example (99 674 PETs), -m 1: BASELINE.json configs[0]", "wall_s": min(walls), "first_wall_s": walls[0],
                "loop_rows": max(0, got.count("\n") - 1), "loop_file_identical_to_golden": bool(same),
                "reference_wall_s": 9.8, "reference_source": "BASELINE.md section 2 (the reference, converted copy, one core, build container)",
                "note": "whole command as a child process: interpreter start + parse + upload + sweep + significance + .loop"}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ---- CPU baseline: the C oracle, one host process per chromosome (the reference's own shape, cLoops/pipe.py:117) ----
def _cpu_worker(job):
    import oracle
    from cloops_amd.synth import synth_chrom
    ci, length, n, runs = job
    X, Y = synth_chrom(n, length, 1000 * CFG + ci)
    d = Y.astype("int64") - X
    out = []
    for eps, m, cut in runs:
        keep = d >= cut
        t0 = time.perf_counter()
        oracle.labels(VARIANT, X[keep], Y[keep], eps, m)
        out.append((time.perf_counter() - t0, int(keep.sum())))
    return out


def reference_python_speed():
    """the reference's own classes in CPython, one core, measured in the build container (the reference never travels to the GPU
    box): profiles/reference_python_speed.json, written by tools/measure_reference_python.py"""
    try:
        with open(os.path.join(ROOT, "profiles", "reference_python_speed.json")) as fh:
            j = json.load(fh)
        return {"reference_python_pets_per_s_per_core": j["reference_python_pets_per_s_per_core"],
                "reference_python_source": "profiles/reference_python_speed.json (%s; %s)" % (j["reference_python_pets_per_s_per_core_note"], j["what"])}
    except Exception:
        return {"reference_python_pets_per_s_per_core": None, "reference_python_source": None}


def cpu_baseline(api, sizes, steps):
    """The CPU oracle (C port of cLoops/cDBSCAN2.py) over the SAME 23 chromosomes, one worker process per chromosome
    up to os.cpu_count() (cLoops/pipe.py:117), on a bounded sample of the sweep: 2 of its 12 runs -- (eps 5000,
    minPts 50, cut 0) and (eps 7500, minPts 30, the GPU chain's cut) -- with the same per-run barrier the reference has
    (runDBSCAN returns when its slowest worker, chr1, is done).  Also a full-size parity check: GPU labels ==
    oracle labels on chr21 at the second setting."""
    import multiprocessing as mp
    import numpy as np
    import oracle
    from cloops_amd.synth import synth_chrom
    oracle.build()
    st = {(s["eps"], s["minPts"]): s for s in steps}
    runs = [(5000, 50, 0), (7500, 30, int(st[(7500, 30)]["cut_in"]))]
    # parity at full size on one chromosome
    ci = 20
    name, length, n = sizes[ci]
    X, Y = synth_chrom(n, length, 1000 * CFG + ci)
    ch = api.Chromosome(X, Y, device=0)
    got = ch.cluster(VARIANT, runs[1][0], runs[1][1], runs[1][2]).labels
    ch.close()
    want = oracle.single_dbscan(VARIANT, X, Y, runs[1][0], runs[1][1], runs[1][2])["labels"]
    same = bool(np.array_equal(got, want))
    workers = max(1, min(os.cpu_count() or 1, len(sizes)))
    jobs = [(k, length, n, runs) for k, (_, length, n) in enumerate(sizes)]
    ctx = mp.get_context("fork")
    with ctx.Pool(workers) as pool:
        per = pool.map(_cpu_worker, jobs, chunksize=1)
    walls = [max(p[r][0] for p in per) for r in range(len(runs))]           # per-run barrier: slowest chromosome
    n_in = [sum(p[r][1] for p in per) for r in range(len(runs))]
    cpu_s = sum(p[r][0] for p in per for r in range(len(runs)))
    return {"value": sum(n_in) / sum(walls), "unit": "PETs/s", "cores": workers, "kind": "port",
            "sample": "2 of the 12 runs of the same sweep over the same 23 chromosomes, %d worker processes (one chromosome each, "
                      "%d host cores visible): %s; %.0f s of CPU work, slowest-chromosome wall %.1f s + %.1f s" % (
                          workers, os.cpu_count() or 1, ", ".join("(eps %d, minPts %d, cut %d)" % r for r in runs), cpu_s, walls[0], walls[1]),
            "single_thread_pets_per_s": sum(p[r][1] for p in per for r in range(len(runs))) / cpu_s,
            "labels_match_gpu": same, "labels_checked_on": "%s (%d PETs) at eps %d minPts %d cut %d" % ((name, n) + runs[1]),
            "note": "kind = port: the sequential C restatement of the reference's classes (oracle/), ~20x faster than the Python reference itself, "
                    "whose own one-core speed is carried in reference_python_pets_per_s_per_core", **reference_python_speed()}


def cpu_baseline_full(n_total, steps, final_cut, ncand):
    """The sequential C oracle over the whole sweep, in the reference's parallel shape (one worker process per chromosome,
    cLoops/pipe.py:117; every cut from the concatenated distance lists through the reference's estimator): the chain runner of
    tests/golden/make_golden_synth200M_chain.py.  Its chain must equal the GPU's (cuts, final cut, candidate count)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import oracle
    oracle.build()
    from make_golden_synth200M_chain import run_chain
    t0 = time.perf_counter()
    res = run_chain(n_total, CFG, MODE3[0], MODE3[1])
    wall = time.perf_counter() - t0
    pets = sum(s["n_in"] for s in res["steps"])
    same = ([s.get("cut_out") for s in res["steps"]] == [s.get("cut_out") for s in steps] and res["final_cut"] == final_cut
            and res["candidates"] == ncand)
    return {"value": pets / res["oracle_wall_s"], "unit": "PETs/s", "cores": min(os.cpu_count() or 1, res["workers"]), "kind": "port",
            "sample": "the WHOLE sweep: 12 runs x 23 chromosomes, %d worker processes (one chromosome each, %d host cores visible), cut chain "
                      "estimated from the concatenated distance lists; %.0f s of CPU work, %.1f s wall for the 12 runs (%.1f s with synthesis)" % (
                          res["workers"], os.cpu_count() or 1, res["oracle_cpu_s"], res["oracle_wall_s"], wall),
            "sweep_wall_s": res["oracle_wall_s"], "chain_matches_gpu": bool(same),
            "note": "kind = port: the sequential C restatement of the reference's classes (oracle/), ~20x faster than the Python reference itself, "
                    "whose own one-core speed is carried in reference_python_pets_per_s_per_core", **reference_python_speed()}


if __name__ == "__main__":
    main()
