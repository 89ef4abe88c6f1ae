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
