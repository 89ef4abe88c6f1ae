"""skch::Map's grouping of reader batches into device passes (mashmap_amd/host/pass_plan.hpp: the hand-over queue and the pass-size ramp),
exercised on the CPU by tests/hostlogic/pass_check.cpp: a producer and a consumer thread under randomised timing -- every item exactly
once, in order, passes never larger than they should be, no deadlock whichever stage is the slow one."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pass_grouping_under_random_timing(tmp_path):
    exe = str(tmp_path / "pass_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "hostlogic", "pass_check.cpp"), "-lpthread"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    lines = p.stdout.splitlines()
    assert p.returncode == 0 and len(lines) == 27 and all(l.startswith("ok ") for l in lines), p.stdout[-2000:]
    by = {l.split()[1]: l.split("sizes ")[1] for l in lines[:7]}
    # the ramp of a 20-batch input whose size is known, passes of at most 4 batches: 1, 1, 2, 4, 4 up, then at most half of what is left
    assert by["fast-producer-known"] == "1,1,2,4,4,4,2,1,1"
    assert by["fast-producer-unknown"] == "1,1,2,4,4,4,4"
    assert by["no-coalescing"] == ",".join(["1"] * 9) and by["single-item"] == "1"


def test_query_batch_plan(tmp_path):
    """skch::queryBatchPlan: 512 Mbp batches and 3 072 Mbp passes per context by default, one batch per pass with several contexts, ASCII
    uploads or MASHMAP_HIP_COALESCE_MBP=0; at most 64 batches per pass; page-locked buffers for one pass queued + one uploading + the
    reader's, never more than the input needs"""
    exe = str(tmp_path / "pass_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "hostlogic", "pass_check.cpp"), "-lpthread"])
    big = str(tmp_path / "q.fa")
    with open(big, "wb") as f:
        f.truncate(20_000_000_000)                                   # a sparse 20 GB "FASTA": only its size is looked at (first bytes: not gzip)
    p = subprocess.run([exe, "plan", big], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    rows = {}
    for l in p.stdout.splitlines():
        f = l.split()
        rows[(f[1], int(f[3]))] = dict(batch=int(f[5]), pas=int(f[7]), buffers=int(f[9]), bufferBytes=int(f[11]), known=int(f[13]))
    d = rows[("default", 1)]
    assert d["batch"] == 512_000_000 and d["pas"] == 3_072_000_000 and d["buffers"] == 14 and d["known"] == 1
    assert 0.375 * 512e6 < d["bufferBytes"] < 0.45 * 512e6            # packed: 3/8 byte per base + slack
    d2 = rows[("default", 2)]
    assert d2["batch"] == 1_024_000_000 and d2["pas"] == d2["batch"] and d2["buffers"] == 8
    assert rows[("coalesce0", 1)]["pas"] == 512_000_000 and rows[("coalesce0", 1)]["buffers"] == 8
    assert rows[("b256c4096", 1)]["batch"] == 256_000_000 and rows[("b256c4096", 1)]["pas"] == 4_096_000_000 and rows[("b256c4096", 1)]["buffers"] == 24
    assert rows[("tiny", 1)]["pas"] == 64 * rows[("tiny", 1)]["batch"]
    a = rows[("ascii", 1)]
    assert a["pas"] == a["batch"] == 512_000_000 and a["bufferBytes"] > 512_000_000
