"""Drop-in boundary, source level: the reference's UNMODIFIED src/map/mash_map.cpp (and the parseCmdArgs.hpp it includes)
must compile and link against this repository's skch::Sketch / skch::Map (mashmap_amd/host/) when
mashmap_amd/host/reference_tree shadows winSketch.hpp and computeMap.hpp.  Only where /root/reference exists."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "map", "mash_map.cpp")), reason="reference tree not present")
def test_reference_main_compiles_against_hip_classes():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mashmap_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    exe = os.path.join(ROOT, "oracle", "_ref", "mashmap_dropin")
    assert os.path.exists(exe)
    p = subprocess.run([exe, "-v"], capture_output=True, text=True)
    assert p.returncode == 0 and "3.1.3" in p.stderr
    # without a GPU the drop-in must fail loudly in skch::Sketch's constructor, not fall back to anything
    import torch
    if not torch.cuda.is_available():
        fa = os.path.join(ROOT, "tests", "golden", "_tiny.fa")
        with open(fa, "w") as f:
            f.write(">a\n" + "ACGT" * 2000 + "\n")
        try:
            p = subprocess.run([exe, "-r", fa, "-q", fa, "-o", "/dev/null"], capture_output=True, text=True)
            assert p.returncode != 0 and "no usable HIP device" in p.stderr
        finally:
            os.remove(fa)


def test_cli_standalone_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    fa = tmp_path / "t.fa"
    fa.write_text(">a\n" + "ACGT" * 2000 + "\n")
    p = subprocess.run([exe, "-r", str(fa), "-q", str(fa), "-o", "/dev/null"], capture_output=True, text=True)
    assert p.returncode != 0 and "no usable HIP device" in p.stderr


def _build_commonfunc_check(out, reftree):
    src = os.path.join(ROOT, "tests", "hostlogic", "commonfunc_check.cpp")
    host = os.path.join(ROOT, "mashmap_amd", "host")
    lib = os.path.join(ROOT, "mashmap_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-w"]
    if reftree:
        cmd += ["-DMASHMAP_HIP_REFERENCE_TREE", "-I" + os.path.join(host, "reference_tree"), "-I" + host, "-I" + os.path.join(REF, "src"),
                "-I" + os.path.join(REF, "src", "common"), "-I" + os.path.join(ROOT, "oracle", "gsl_shim")]
    cmd += ["-o", out, src, "-L" + lib, "-lmashmap_hip", "-Wl,-rpath," + lib, "-lz", "-lpthread"]
    subprocess.check_call(cmd)
    return out


@pytest.mark.parametrize("reftree", [False, True])
def test_commonfunc_level_callers_compile(tmp_path, reftree):
    """SURVEY section 8b's inner seams: a caller of skch::CommonFunc::sketchSequence / addMinmers (commonFunc.hpp:183, :302) compiles
    against mashmap_amd/host/skch_commonfunc.hpp, standalone and overlaid on the reference's own commonFunc.hpp"""
    if reftree and not os.path.exists(os.path.join(REF, "src", "map", "include", "commonFunc.hpp")):
        pytest.skip("reference tree not present")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mashmap_amd", "csrc")])
    exe = _build_commonfunc_check(str(tmp_path / "cfc"), reftree)
    import torch
    if not torch.cuda.is_available():
        seqs = tmp_path / "s.txt"
        seqs.write_text("ACGT" * 100 + "\n")
        p = subprocess.run([exe, "19", "200", "10", str(seqs)], capture_output=True, text=True)
        assert p.returncode != 0 and "no usable HIP device" in p.stderr      # no CPU fallback behind the seam either
