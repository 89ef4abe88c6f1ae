"""bench.py's rank plumbing on CPU: `--gpus N` must start N ranks itself and may never print an n_gpus other than N.
Runs with --stub (gloo, no kernels: the JSON says "data": "stub"), because there is no GPU here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return e


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub"], capture_output=True, text=True, env=_env(), timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["data"] == "stub"


def test_gpus_1_runs_in_process():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0", "--stub"], capture_output=True, text=True, env=_env(), timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _last_json(p.stdout)["n_gpus"] == 1


def test_refuses_a_world_size_that_differs_from_gpus():
    e = _env(); e.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub"], capture_output=True, text=True, env=e, timeout=120)
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_launched_by_torchrun_with_matching_world_size():
    """the driver's own form: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--stub"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _last_json(p.stdout)["n_gpus"] == 2
