"""CPU: `python bench.py --gpus N` started plainly (no torch.distributed.run) must be its own launcher -- N fresh
ranks with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rank 0's JSON line on stdout, a failing rank fails the run.
--dry-launch swaps the GPU body of a rank for a gloo rendezvous on the CPU (the stub device check), so the launcher
itself -- environment, ports, output routing, exit codes -- is what runs here.  BASELINE configs[4] (8 GPUs) is this
launcher + the per-rank step tests/test_c5_gpu.py covers at full per-rank size."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus2_without_torchrun_spawns_two_ranks():
    p = _run(["--gpus", "2", "--dry-launch"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    r = json.loads(lines[0])
    assert r == {"launcher": "dry", "n_gpus": 2, "rank_sum": 1, "local_rank_sum": 1, "ranks": 2}


def test_gpus4_ranks_are_distinct():
    p = _run(["--gpus", "4", "--dry-launch"])
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert (r["n_gpus"], r["rank_sum"], r["local_rank_sum"], r["ranks"]) == (4, 6, 6, 4)


def test_a_failing_rank_fails_the_run_and_stops_the_others():
    p = _run(["--gpus", "2", "--dry-launch"], {"CLAP_BENCH_DRY_FAIL_RANK": "1"})
    assert p.returncode == 3
    assert "rank 1 exited with 3" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")], "no result line from a failed run"


def test_under_a_launcher_environment_no_second_spawn():
    """With WORLD_SIZE already set (torch.distributed.run, the driver's N > 1 form) bench.py is one rank."""
    from bench import _free_port
    p = _run(["--gpus", "1", "--dry-launch"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                                                "MASTER_PORT": str(_free_port())})
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["ranks"] == 1


def test_more_gpus_than_devices_is_refused_before_any_spawn():
    import torch
    have = torch.cuda.device_count()
    p = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert p.returncode == 2 and "exposes" in p.stderr
