"""End to end: the `mashmap_hip` command line (mashmap_amd/host/: skch::Sketch + skch::Map on the C ABI) must write the same
PAF, byte for byte, as the reference binary for the same arguments.  Compared against the committed fixtures
(tests/golden/paf/, produced by the real reference) and, where oracle/_ref/mashmap_ref travelled along, against a live run."""
import os
import subprocess

import pytest

import mmutil as U
from golden import cases as CS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_BIN = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
DROPIN_BIN = os.path.join(ROOT, "oracle", "_ref", "mashmap_dropin")
PAF_DIR = os.path.join(ROOT, "tests", "golden", "paf")
CASES = {c[0]: c for c in CS.paf_cases()}


def _run(binary, td, name, refrec, qrec, extra, tag, threads="4"):
    rf = os.path.join(td, name + ".ref.fa")
    if not os.path.exists(rf):
        U.write_fasta(rf, refrec)
    out = os.path.join(td, "%s.%s.paf" % (name, tag))
    args = [binary, "-r", rf, "-o", out, "-t", threads] + extra
    if qrec is not None:
        qf = os.path.join(td, name + ".q.fa")
        if not os.path.exists(qf):
            U.write_fasta(qf, qrec)
        args += ["-q", qf]
    p = subprocess.run(args, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return open(out, "rb").read()


def _diff(a, b):
    la, lb = a.decode().splitlines(), b.decode().splitlines()
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d:\n  got %s\n  exp %s\n(%d vs %d lines)" % (i, x, y, len(la), len(lb))
    return "%d vs %d lines; first extra: %s" % (len(la), len(lb), (la[len(lb):] or lb[len(la):])[:1])


@pytest.mark.parametrize("name", sorted(CASES))
def test_paf_identical_to_reference(name, tmp_path):
    assert os.path.exists(HIP_BIN), "mashmap_hip not built (python -c 'import __graft_entry__ as g; g.build()')"
    _, refrec, qrec, extra = CASES[name]
    got = _run(HIP_BIN, str(tmp_path), name, refrec, qrec, extra, "hip")
    exp = open(os.path.join(PAF_DIR, name + ".paf"), "rb").read()
    assert len(exp) > 0
    assert got == exp, _diff(got, exp)
    if os.path.exists(U.REF_BIN):
        live = _run(U.REF_BIN, str(tmp_path), name, refrec, qrec, extra, "ref")
        assert got == live, _diff(got, live)
    if os.path.exists(DROPIN_BIN):
        # the reference's unmodified main() + option parser on top of this repository's skch::Sketch / skch::Map
        dr = _run(DROPIN_BIN, str(tmp_path), name, refrec, qrec, extra, "dropin")
        assert dr == exp, _diff(dr, exp)


def test_paf_independent_of_batching_and_threads(tmp_path):
    _, refrec, qrec, extra = CASES["default"]
    a = _run(HIP_BIN, str(tmp_path), "default", refrec, qrec, extra, "t1", threads="1")
    os.environ["MASHMAP_HIP_BATCH_MBP"] = "0.05"          # ~5 reads per device pass
    try:
        b = _run(HIP_BIN, str(tmp_path), "default", refrec, qrec, extra, "t7", threads="7")
    finally:
        del os.environ["MASHMAP_HIP_BATCH_MBP"]
    assert a == b


def test_paf_from_gzipped_fastq_queries(tmp_path):
    """the host reader (mashmap_amd/host/seq_reader.hpp): FASTQ and gzip give the same PAF as the FASTA of the same reads"""
    import gzip
    _, refrec, qrec, extra = CASES["default"]
    rf = str(tmp_path / "ref.fa.gz")
    with gzip.open(rf, "wb") as f:
        for n, a in refrec:
            f.write(b">" + n.encode() + b" some description\n" + a.tobytes() + b"\n")
    qf = str(tmp_path / "q.fq.gz")
    with gzip.open(qf, "wb") as f:
        for n, a in qrec:
            f.write(b"@" + n.encode() + b" extra words\n" + a.tobytes() + b"\n+\n" + b"I" * len(a) + b"\n")
    out = str(tmp_path / "o.paf")
    # the reference derives the sketch size from the reference FILE size (compressed here): pin it to what the fixture used
    p = subprocess.run([HIP_BIN, "-r", rf, "-q", qf, "-o", out, "-t", "3", "-J", str(_fixture_sketch_size())], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert open(out, "rb").read() == open(os.path.join(PAF_DIR, "default.paf"), "rb").read()


def _fixture_sketch_size():
    """sketch size mashmap derives for the uncompressed FASTA of the 'default' case (what produced tests/golden/paf/default.paf)"""
    from mashmap_amd import capi
    _, refrec, _, _ = CASES["default"]
    nbytes = sum(len(n) + 2 + len(a) + (len(a) + 79) // 80 for n, a in refrec)        # U.write_fasta: 80 columns
    return int(capi.load().mm_stat_recommended_sketch_size(19, 0.85, 5000, nbytes))
