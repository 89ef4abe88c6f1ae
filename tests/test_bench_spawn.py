"""bench.py's rank plumbing on CPU: `--gpus N` must start N ranks itself and may never print an n_gpus other than N.
Runs with --stub (gloo, no kernels: the JSON says "data": "stub"), because there is no GPU here."""
import json
import os
import subprocess
import sys

import pytest

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


# ---- the rank loop itself (bench.StepLoop: the code a real N-GPU run executes), driven over gloo with the stub context's event log
def _events(prefix, world):
    return [[json.loads(l) for l in open("%s.%d" % (prefix, r))] for r in range(world)]


@pytest.mark.parametrize("world", [2, 4, 8])       # 8: the node the driver's scaling run uses
def test_rank_loop_overlapped_exchange_order_and_timed_region(world, tmp_path):
    """step = rotate batch, map, _end of the previous exchange, _begin of this one; the fence before the clock starts leaves no exchange in
    flight, the last exchange of the timed steps is waited for BEFORE the clock stops; one exchange in flight at most; every exchange
    carries the records of the pass it was begun after, rank-major with the per-rank counts; three resident batches take turns"""
    steps, warmup = 5, 2
    e = _env(); e["MM_STUB_LOG"] = str(tmp_path / "ev")
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup), "--stub"], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == world and d["passes"]["timed"] == steps and d["passes"]["resident_batches"] == 3
    for r, ev in enumerate(_events(e["MM_STUB_LOG"], world)):
        kinds = [x["ev"] for x in ev]
        assert kinds.count("map") == steps + warmup and kinds.count("begin") == steps + warmup and kinds.count("end") == steps + warmup
        assert [x["slot"] for x in ev if x["ev"] == "exchange"] == [i % 2 for i in range(steps + warmup)]
        t0, t1 = kinds.index("t0"), kinds.index("t1")
        # nothing in flight across either clock edge
        depth = 0
        for i, k in enumerate(kinds):
            depth += k == "begin"; depth -= k == "end"
            assert 0 <= depth <= 1
            if i in (t0, t1):
                assert depth == 0, "an exchange is in flight when the clock %s" % ("starts" if i == t0 else "stops")
        timed = kinds[t0:t1]
        assert timed.count("map") == steps and timed.count("begin") == steps and timed.count("end") == steps
        # the exchange ended in step i+1 (or in the fence) is the one begun after pass i, with every rank's count in rank order
        ends = [x for x in ev if x["ev"] == "end"]
        assert [x["of"] for x in ends] == list(range(1, steps + warmup + 1))
        assert all(x["counts"] == [3 + q for q in range(world)] for x in ends)
    assert d["ms_per_step"] >= 2.0                                   # max over ranks of at least the stub's 2 ms per pass


def test_rank_loop_with_a_rank_that_has_no_mappings(tmp_path):
    e = _env(); e["MM_STUB_LOG"] = str(tmp_path / "ev"); e["MM_STUB_EMPTY_RANK"] = "1"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"], capture_output=True, text=True, env=e, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _last_json(p.stdout)["n_gpus"] == 2
    for ev in _events(e["MM_STUB_LOG"], 2):
        assert all(x["counts"] == [3, 0] for x in ev if x["ev"] == "end")


def test_rank_loop_sync_exchange(tmp_path):
    e = _env(); e["MM_STUB_LOG"] = str(tmp_path / "ev")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub", "--sync-exchange"], capture_output=True, text=True, env=e, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    for ev in _events(e["MM_STUB_LOG"], 2):
        kinds = [x["ev"] for x in ev]
        assert kinds.count("sync_gather") == 3 and "begin" not in kinds


def test_rank_loop_a_rank_failing_in_its_exchange_ends_the_run_without_a_line(tmp_path):
    """rank 1 raises in its third _end: the launcher must come back with a non-zero exit code (the other rank is torn down, not left
    hanging in the collective) and no JSON line may be printed"""
    e = _env(); e["MM_STUB_FAIL_END"] = "1:3"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub"], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode != 0
    assert "stub: rank 1 fails in its exchange 3" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
