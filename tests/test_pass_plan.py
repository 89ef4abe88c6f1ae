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
    # the ramp of a 20-batch input whose size is known (what the e2e run of bench.py shows: 1, 1, 2, 4, 4 batches up, 3, 2, 1, 1, 1 down)
    assert by["fast-producer-known"] == "1,1,2,4,4,3,2,1,1,1"
    assert by["fast-producer-unknown"] == "1,1,2,4,4,4,4"
    assert by["no-coalescing"] == ",".join(["1"] * 9) and by["single-item"] == "1"
