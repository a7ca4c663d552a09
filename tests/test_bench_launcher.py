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
    shape = {k: r.pop(k) for k in ("line_keys", "c5_keys")}
    assert r == {"launcher": "dry", "n_gpus": 2, "rank_sum": 1, "local_rank_sum": 1, "ranks": 2,
                 "rccl_ranks": 2, "devices": ["cpu:0", "cpu:1"]}      # the line proves its own N: one device per rank
    assert "c5" in shape["line_keys"] and shape["c5_keys"]


def test_gpus8_line_has_the_configs4_leg_beside_the_weak_scaling_headline():
    """What the driver runs on an 8-GPU node is plain `bench.py --gpus 8`: the headline stays BASELINE configs[1] per GPU
    (comparable with N = 1: weak scaling), and the same run times configs[4]'s per-rank share -- 16 M entities + 2 M
    particles in all -- as a second leg reported under `c5`; `summary` and `c5` come before the long sections.  An
    explicit sizing (--c5, --chains ...) is one leg, as asked."""
    p = _run(["--gpus", "8", "--dry-launch"], timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 8 and r["ranks"] == 8 and r["rccl_ranks"] == 8 and len(set(r["devices"])) == 8
    keys = r["line_keys"]
    assert keys.index("summary") < keys.index("c5") < keys.index("config") < keys.index("roofline")
    assert {"value", "ms_per_step", "entities_per_gpu", "particles_per_gpu", "visible", "particle_updates_per_s"} <= set(r["c5_keys"])
    p = _run(["--gpus", "2", "--dry-launch", "--c5"])
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["c5_keys"] is None and "c5" not in r["line_keys"]


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


def test_two_ranks_on_one_device_are_refused():
    """The proof-of-participation check the GPU run ends its set-up on (bench.check_proof: the communicator's own rank
    count == --gpus, every rank's communicator rank == its launcher rank, N distinct PCI bus ids), driven here with the
    gloo stand-in: two ranks reporting the same device -> no JSON line, exit code 5."""
    p = _run(["--gpus", "2", "--dry-launch"], {"CLAP_BENCH_DRY_SAME_DEVICE": "1"})
    assert p.returncode == 5, (p.returncode, p.stderr[-1500:])
    assert "refusing to report" in p.stderr and "distinct device" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_check_proof_cases():
    from bench import check_proof
    ok = dict(rccl_ranks=4, comm_rank_ok=True, devices=["0000:05:00.0", "0000:15:00.0", "0000:65:00.0", "0000:75:00.0"])
    assert check_proof(ok, 4) is None
    assert "communicator has 1 ranks" in check_proof(dict(ok, rccl_ranks=1), 4)        # a world of one, N times
    assert "launcher rank" in check_proof(dict(ok, comm_rank_ok=False), 4)
    assert "3 distinct" in check_proof(dict(ok, devices=ok["devices"][:3] + ok["devices"][:1]), 4)


def test_a_rank_that_never_arrives_is_ended_by_its_watchdog():
    """ncclCommInitRank / the rendezvous block for ever when a peer never arrives: every rank arms a watchdog around its
    collective set-up (bench.InitWatchdog) that ends the rank with exit code 4; the launcher stops the others."""
    p = _run(["--gpus", "2", "--dry-launch"], {"CLAP_BENCH_DRY_HANG_RANK": "1", "CLAP_BENCH_DRY_INIT_TIMEOUT": "6"}, timeout=120)
    assert p.returncode == 4, (p.returncode, p.stderr[-1500:])
    assert "did not finish within 6 s" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


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
