"""CPU: bench.py's launch contract -- `--gpus N` really starts N ranks (gloo here, RCCL on the GPU box), the 23
chromosomes are LPT-sharded over them, the chained cut comes out of the all-reduced statistics and equals the
single-process chain.  The GPU handle is replaced by the oracle-backed stand-in (tests/bench_cpu_hook.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(gpus):
    env = dict(os.environ)
    env["CLOOPS_BENCH_PRELOAD"] = "bench_cpu_hook"
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, env.get("PYTHONPATH", "")])
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--backend", "gloo",
                          "--n-total", "400000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly one JSON line on stdout
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_two_ranks_and_matches_single_process():
    one = _bench(1)
    two = _bench(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for j in (one, two):
        assert "configs[3]" in j["config"]["workload"] and j["config"]["chromosomes"] == 23
        assert j["config"]["runs_per_sweep"] == 12 and j["scaling"] == "strong" and j["unit"] == "PETs/s"
    assert any(c is not None for c in one["config"]["cuts"])
    assert two["config"]["cuts"] == one["config"]["cuts"]
    assert two["config"]["final_cut"] == one["config"]["final_cut"]
    assert two["config"]["candidate_loops"] == one["config"]["candidate_loops"]
    assert two["config"]["pets_entering_dbscan_per_sweep"] == one["config"]["pets_entering_dbscan_per_sweep"]


def _preflight_world2(children, timeout):
    """bench.rccl_preflight on two 'ranks' of one launch (two threads: same parent process, i.e. the same launch tag), each
    with a stand-in for the child that would try RCCL -> [(ok, note), (ok, note)]"""
    import threading
    sys.path.insert(0, ROOT)
    import bench
    out = [None, None]

    def run(rank):
        out[rank] = bench.rccl_preflight(rank, 2, rank, timeout=timeout, child=children[rank])
    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    return out


def test_preflight_all_ranks_take_the_same_decision(monkeypatch):
    """the RCCL-direct pre-flight of bench.py: whatever happens to one rank's child (error, hang), every rank reads the same
    verdict -- the ranks cannot split between libcloops_comm.so and torch.distributed"""
    monkeypatch.setenv("MASTER_PORT", "1")
    ok = "import sys"
    bad = "import sys; sys.stderr.write('hipIpcGetMemHandle: invalid argument'); sys.exit(3)"
    hang = "import time; time.sleep(60)"
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "pf-a")
    a = _preflight_world2([ok, ok], 20.0)
    assert a == [(True, ""), (True, "")], a
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "pf-b")
    b = _preflight_world2([ok, bad], 20.0)
    assert b[0][0] is False and b[1][0] is False and b[0] == b[1] and "rank 1: exit 3" in b[0][1] and "hipIpcGetMemHandle" in b[0][1], b
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "pf-c")
    c = _preflight_world2([hang, ok], 2.0)
    assert c[0][0] is False and c[1][0] is False and c[0] == c[1] and "rank 0: no answer" in c[0][1], c
