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


@pytest.mark.parametrize("name", [c[0] for c in CS.paf_list_cases()])
def test_paf_with_file_lists(name, tmp_path):
    """--rl / --ql: several reference files share one seqId space (winSketch.hpp:174-214), several query files one read counter; the
    configs[4] form (--dense --pi 80, 20 kbp reads at 15-20 % error, ten reference files)"""
    _, ref_files, q_files, extra = {c[0]: c for c in CS.paf_list_cases()}[name]
    td = str(tmp_path)
    rl, ql = os.path.join(td, "refs.txt"), os.path.join(td, "queries.txt")
    with open(rl, "w") as f:
        for i, recs in enumerate(ref_files):
            fn = os.path.join(td, "ref%d.fa" % i); U.write_fasta(fn, recs); f.write(fn + "\n")
    with open(ql, "w") as f:
        for i, recs in enumerate(q_files):
            fn = os.path.join(td, "q%d.fa" % i); U.write_fasta(fn, recs); f.write(fn + "\n")
    exp = open(os.path.join(PAF_DIR, name + ".paf"), "rb").read()
    assert len(exp) > 0
    for binary, tag in ((HIP_BIN, "hip"), (DROPIN_BIN, "dropin"), (U.REF_BIN, "ref")):
        if tag != "hip" and not os.path.exists(binary):
            continue
        out = os.path.join(td, tag + ".paf")
        p = subprocess.run([binary, "--rl", rl, "--ql", ql, "-o", out, "-t", "4"] + extra, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        got = open(out, "rb").read()
        assert got == exp, tag + ": " + _diff(got, exp)
    env = dict(os.environ, MASHMAP_HIP_DEVICES="0,0")      # and sharded over two contexts
    out = os.path.join(td, "sharded.paf")
    p = subprocess.run([HIP_BIN, "--rl", rl, "--ql", ql, "-o", out, "-t", "4"] + extra, capture_output=True, text=True, env=env)
    assert p.returncode == 0 and open(out, "rb").read() == exp


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
    """the host reader (mashmap_amd/host/seq_parse.hpp): FASTQ and gzip give the same PAF as the FASTA of the same reads"""
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


def _read_index_files(prefix):
    import numpy as np
    mdt = np.dtype([("hash", "<u8"), ("wpos", "<i4"), ("wpos_end", "<i4"), ("seqId", "<i4"), ("strand", "<i2"), ("pad", "<i2")])
    pdt = np.dtype([("pos", "<i4"), ("pad0", "<i4"), ("hash", "<u8"), ("seqId", "<i4"), ("side", "i1"), ("pad1", "i1", (3,))])
    raw = open(prefix + ".index", "rb").read()
    n = int(np.frombuffer(raw[:8], dtype="<u8")[0])
    mins = np.frombuffer(raw[8:], dtype=mdt, count=n)
    raw = open(prefix + ".map", "rb").read()
    nk = int(np.frombuffer(raw[:8], dtype="<u8")[0])
    off, keys, lists = 8, [], []
    for _ in range(nk):
        key, cnt = np.frombuffer(raw[off:off + 16], dtype="<u8")
        off += 16
        pts = np.frombuffer(raw[off:off + 24 * int(cnt)], dtype=pdt)
        off += 24 * int(cnt)
        keys.append(int(key)); lists.append([(int(p["pos"]), int(p["hash"]), int(p["seqId"]), int(p["side"])) for p in pts])
    assert off == len(raw)
    return [tuple(int(m[f]) for f in ("hash", "wpos", "wpos_end", "seqId", "strand")) for m in mins], keys, lists


def test_save_and_load_index_interoperate_with_the_reference(tmp_path):
    """--saveIndex / --loadIndex in the reference's on-disk layout (winSketch.hpp:284-374): same records, same map, same key order;
    each program maps from the other's files and still writes the golden PAF"""
    _, refrec, qrec, extra = CASES["default"]
    td = str(tmp_path)
    exp = open(os.path.join(PAF_DIR, "default.paf"), "rb").read()
    got = _run(HIP_BIN, td, "default", refrec, qrec, ["--saveIndex", td + "/hipidx"], "hsave")
    assert got == exp
    mine = _read_index_files(td + "/hipidx")
    assert len(mine[0]) > 1000 and len(mine[1]) > 500
    again = _run(HIP_BIN, td, "default", refrec, qrec, ["--loadIndex", td + "/hipidx"], "hload")
    assert again == exp
    tsv = _run(HIP_BIN, td, "default", refrec, qrec, ["--saveIndex", td + "/hip.tsv"], "htsv")
    assert tsv == exp and open(td + "/hip.tsv").readline().split("\t")[0] == "seqId"
    assert _run(HIP_BIN, td, "default", refrec, qrec, ["--loadIndex", td + "/hip.tsv"], "htsvload") == exp
    if os.path.exists(U.REF_BIN):
        assert _run(U.REF_BIN, td, "default", refrec, qrec, ["--saveIndex", td + "/refidx"], "rsave") == exp
        theirs = _read_index_files(td + "/refidx")
        assert mine[0] == theirs[0], "minmerIndex on disk differs"
        assert mine[1] == theirs[1], "lookup keys (or their order) differ"
        assert mine[2] == theirs[2], "interval point lists differ"
        assert _run(HIP_BIN, td, "default", refrec, qrec, ["--loadIndex", td + "/refidx"], "hloadref") == exp
        assert _run(U.REF_BIN, td, "default", refrec, qrec, ["--loadIndex", td + "/hipidx"], "rloadhip") == exp


def test_contig_beyond_int32_is_refused_not_truncated(tmp_path):
    """offset_t is int32 here as in the reference's default build (base_types.hpp:17-22); its -DLARGE_CONTIG variant (64-bit
    coordinates, CMakeLists.txt:23) is not provided.  A 2^31 + 1 bp contig -- as reference or as query -- must end the run with exit
    code 1 and a message in the reference's style that names the sequence, not wrap its length into a negative int."""
    big = os.path.join(str(tmp_path), "big.fa")
    n = (1 << 31) + 1
    with open(big, "wb") as f:
        f.write(b">giant\n")
        line = b"ACGTTGCA" * 8192 + b"\n"                      # 65 536 bases per line
        full, rest = divmod(n, 65536)
        blk = line * 256
        for _ in range(full // 256):
            f.write(blk)
        f.write(line * (full % 256))
        f.write(b"A" * rest + b"\n")
    small = os.path.join(str(tmp_path), "small.fa")
    U.write_fasta(small, [("chr0", U.random_dna(7, 60000))])
    out = os.path.join(str(tmp_path), "o.paf")
    for ref, qry, where in ((big, small, "Sketch::build] ERROR: reference sequence giant has 2147483649 bp"),
                            (small, big, "Map::mapQuery] ERROR: query sequence giant has 2147483649 bp")):
        p = subprocess.run([HIP_BIN, "-r", ref, "-q", qry, "-o", out, "-t", "4"], capture_output=True, text=True)
        assert p.returncode == 1, (p.returncode, p.stderr[-600:])
        assert where in p.stderr and "LARGE_CONTIG" in p.stderr, p.stderr[-600:]
