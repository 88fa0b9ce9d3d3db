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
