"""mm_heap.h (used by k_l2_select on the device) must move elements exactly as libstdc++'s std::make_heap / std::pop_heap, which is
what orders a fragment's L1 candidates in the reference (computeMap.hpp:791,1256).  Compiled for the host and checked here."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_heap_matches_libstdcxx(tmp_path):
    exe = str(tmp_path / "heap_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "hostlogic", "heap_check.cpp")])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0 and "identical" in p.stdout, p.stdout + p.stderr
