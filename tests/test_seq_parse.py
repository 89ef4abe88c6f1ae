"""mashmap_amd/host/seq_parse.hpp (multi-threaded FASTA / FASTQ / gzip / BGZF ingest) against a plain Python restatement of
seqiter::for_each_seq_in_file's record semantics (src/common/seqiter.hpp:20-111).  CPU only."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import mmutil as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("parse") / "parse_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", out, os.path.join(ROOT, "tests", "hostlogic", "parse_check.cpp"), "-lz", "-lpthread"])
    return out


def fnv(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def records():
    rs = []
    rng = np.random.default_rng(5)
    for i in range(400):
        n = int(rng.choice([0, 1, 17, 300, 5000, 12345, 70000]))
        a = U.random_dna(1000 + i, max(n, 1))[:n]
        if i % 7 == 0 and n:
            a = U.lowercase_some(a, i)
        hdr = "rec%d" % i + (" description words" if i % 3 == 0 else "") + ("\tTAB" if i % 11 == 0 else "")
        rs.append((hdr, a.tobytes()))
    rs.append(("big_one", U.random_dna(77, 900000).tobytes()))
    rs.append(("@looks_like_fastq", b"ACGT" * 10))
    return rs


def fasta_bytes(rs, width):
    out = bytearray()
    for hdr, seq in rs:
        out += b">" + hdr.encode() + b"\n"
        if width:
            for i in range(0, len(seq), width):
                out += seq[i:i + width] + b"\n"
        else:
            out += seq + b"\n"
    return bytes(out)


def fastq_bytes(rs):
    out = bytearray()
    for i, (hdr, seq) in enumerate(rs):
        q = (b"@" if i % 5 == 0 else b"I") + b"+" * max(0, len(seq) - 1) if seq else b""     # quality lines that start with '@' / hold '+'
        out += b"@" + hdr.encode() + b"\n" + seq + b"\n+\n" + q[:len(seq)] + b"\n"
    return bytes(out)


def bgzf_bytes(raw, block=60000):
    out = bytearray()
    for i in list(range(0, len(raw), block)) + [None]:
        chunk = b"" if i is None else raw[i:i + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        data = c.compress(chunk) + c.flush()
        bsize = len(data) + 25
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize) + data + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


def expect(rs, prefix=""):
    lines = []
    for hdr, seq in rs:
        name = hdr.split(" ")[0]
        keep = name.startswith(prefix)
        s = seq if keep else b""
        lines.append("%s\t%d\t%d" % (name, len(s), fnv(s)))
    return lines


@pytest.mark.parametrize("kind", ["fasta1", "fasta60", "fastq", "fasta60.gz", "fastq.gz", "fasta60.bgzf", "fastq.bgzf"])
def test_parallel_reader_matches_the_serial_semantics(exe, tmp_path, kind):
    rs = records()
    raw = fastq_bytes(rs) if kind.startswith("fastq") else fasta_bytes(rs, 60 if "60" in kind else 0)
    path = str(tmp_path / ("in." + kind))
    if kind.endswith(".gz"):
        with gzip.open(path, "wb") as f:
            f.write(raw)
    elif kind.endswith(".bgzf"):
        open(path, "wb").write(bgzf_bytes(raw))
    else:
        open(path, "wb").write(raw)
    want = expect(rs)
    for window, threads in ((1 << 40, 1), (70000, 3), (65536, 8), (300000, 5)):
        p = subprocess.run([exe, str(window), str(threads), path], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        got = p.stdout.splitlines()
        assert got == want, (kind, window, threads, len(got), len(want), [x for x in zip(got, want) if x[0] != x[1]][:3])
    if kind == "fasta60":
        p = subprocess.run([exe, "100000", "4", "--prefix", "rec1", path], capture_output=True, text=True)
        assert p.stdout.splitlines() == expect(rs, "rec1")                    # filtered records are still reported, empty (seqiter.hpp:84-97)
        # a file without a final line break, and two files in a row
        open(path + ".nonl", "wb").write(raw.rstrip(b"\n"))
        p = subprocess.run([exe, "90000", "3", path + ".nonl", path], capture_output=True, text=True)
        assert p.stdout.splitlines() == want + want


@pytest.mark.parametrize("threads", [1, 2, 7, 32])
def test_worker_pool_runs_every_task_exactly_once(exe, threads):
    """mmhost::WorkerPool (the persistent workers of the reader and the post stage): 4000 back-to-back runs of 1 .. 4096 tasks"""
    out = subprocess.run([exe, "pool", str(threads), "4000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("pool ok 4000"), out.stdout + out.stderr


@pytest.mark.parametrize("kind", ["fasta60", "fastq", "fasta60.gz", "fasta60.bgzf"])
def test_non_regular_inputs_are_read_as_streams(exe, tmp_path, kind):
    """-q <(cat x.fa) / <(gzip -c x.fa) / a FIFO: the reference's igzstream reads them; no mmap, no sniffing handle that eats bytes"""
    rs = records()[:120]
    raw = fastq_bytes(rs) if kind.startswith("fastq") else fasta_bytes(rs, 60)
    data = gzip.compress(raw) if kind.endswith(".gz") else bgzf_bytes(raw) if kind.endswith(".bgzf") else raw
    path = str(tmp_path / "plain.in")
    open(path, "wb").write(data)
    want = expect(rs)
    for window, threads in ((70000, 3), (1 << 40, 1)):
        p = subprocess.run(["bash", "-c", '"%s" %d %d <(cat "%s")' % (exe, window, threads, path)], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert p.stdout.splitlines() == want
    fifo = str(tmp_path / "fifo")
    os.mkfifo(fifo)
    feeder = subprocess.Popen(["bash", "-c", 'cat "%s" > "%s"' % (path, fifo)])
    p = subprocess.run([exe, "65536", "4", fifo], capture_output=True, text=True, timeout=120)
    feeder.wait()
    assert p.returncode == 0 and p.stdout.splitlines() == want, p.stderr


def test_truncated_or_corrupt_gzip_is_an_error(exe, tmp_path):
    rs = records()[:200]
    raw = fasta_bytes(rs, 60)
    gz = gzip.compress(raw)
    cut = str(tmp_path / "cut.fa.gz")
    open(cut, "wb").write(gz[:len(gz) * 2 // 3])
    p = subprocess.run([exe, "70000", "3", cut], capture_output=True, text=True)
    assert p.returncode != 0 and "inflating" in p.stderr, (p.returncode, p.stderr[-200:])
    bg = bytearray(bgzf_bytes(raw))
    bg[len(bg) // 2] ^= 0x55                                   # a flipped payload byte: inflate error or CRC mismatch
    bad = str(tmp_path / "bad.fa.bgzf")
    open(bad, "wb").write(bytes(bg))
    p = subprocess.run([exe, "70000", "3", bad], capture_output=True, text=True)
    assert p.returncode != 0 and "BGZF" in p.stderr, (p.returncode, p.stderr[-200:])


def _normalise(seq):
    t = bytearray(b"N" * 256)
    for c in b"ACGT":
        t[c] = c; t[c + 32] = c
    return bytes(seq).translate(bytes(t))


@pytest.mark.parametrize("kind", ["fasta60", "fastq", "fasta60.gz"])
def test_packing_reader_encodes_the_normalised_sequences(exe, tmp_path, kind):
    """BatchReader with packOutput (what skch::Map feeds mm_reads_upload_packed): the 2-bit codes + N mask of every record decode to
    makeUpperCaseAndValidDNA of its sequence (commonFunc.hpp:97), records start on 32-base boundaries, padding is zero"""
    rs = records()
    rng = np.random.default_rng(9)
    rs = [(h, (bytes(rng.choice(np.frombuffer(b"ACGTNnacgtRYKM*-", dtype=np.uint8), len(s))) if i % 9 == 4 else s)) for i, (h, s) in enumerate(rs)]
    raw = fastq_bytes(rs) if kind.startswith("fastq") else fasta_bytes(rs, 60)
    path = str(tmp_path / ("in." + kind))
    if kind.endswith(".gz"):
        with gzip.open(path, "wb") as f:
            f.write(raw)
    else:
        open(path, "wb").write(raw)
    want = ["%s\t%d\t%d" % (h.split(" ")[0], len(s), fnv(_normalise(s))) for h, s in rs]
    for window, threads in ((1 << 40, 1), (70000, 3), (300000, 8)):
        p = subprocess.run([exe, str(window), str(threads), "--packed", path], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout[-300:] + p.stderr
        got = p.stdout.splitlines()
        assert got == want, (kind, window, threads, [x for x in zip(got, want) if x[0] != x[1]][:3])


def test_packing_reader_falls_back_on_near_empty_records(exe, tmp_path):
    """thousands of one-base records: their packed form (32 bases each) is larger than their bytes in the file, the single-pass packer's
    regions overflow and the window is redone by the two-pass path -- same records either way"""
    rs = [("t%d" % i, bytes([b"ACGTN"[i % 5]]) * (1 + i % 3)) for i in range(60000)] + [("big", U.random_dna(5, 200000).tobytes())]
    path = str(tmp_path / "tiny.fa")
    open(path, "wb").write(fasta_bytes(rs, 0))
    want = ["%s\t%d\t%d" % (h, len(s), fnv(_normalise(s))) for h, s in rs]
    for window, threads in ((1 << 40, 4), (200000, 8)):
        p = subprocess.run([exe, str(window), str(threads), "--packed", path], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout[-300:] + p.stderr
        assert p.stdout.splitlines() == want


def test_available_cpus_follows_override_affinity_and_quota(exe):
    """the number the reader / post stage widths are capped by (seq_parse.hpp availableCpus): MASHMAP_HIP_CPUS wins; otherwise no more
    than the hardware threads, the affinity mask, or a cgroup CPU quota if this container has one"""
    env = dict(os.environ); env.pop("MASHMAP_HIP_CPUS", None)
    n = int(subprocess.check_output([exe, "cpus"], env=env).split()[0])
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            assert n <= max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    assert int(subprocess.check_output([exe, "cpus"], env=dict(env, MASHMAP_HIP_CPUS="3")).split()[0]) == 3
    if hasattr(os, "sched_setaffinity") and len(os.sched_getaffinity(0)) > 1:
        one = sorted(os.sched_getaffinity(0))[:1]
        out = subprocess.check_output(["python3", "-c", "import os,subprocess,sys; os.sched_setaffinity(0, {%d}); sys.stdout.write(subprocess.check_output([%r, 'cpus']).decode())" % (one[0], exe)], env=env)
        assert int(out.split()[0]) == 1


@pytest.mark.parametrize("width", [1, 7, 31, 32, 33, 60, 61, 0])
def test_packing_reader_splits_long_records_across_threads(exe, tmp_path, width):
    """records longer than a thread's piece of the window (contigs, chromosomes): the packed parser cuts INSIDE them, at line boundaries
    (parseWindowPackedSplit) -- parts that do not start on a 32-base boundary of their record hand their first bases to the thread before
    them.  Line widths around 32, tiny records and empty records between the long ones, '\\r', blank lines, N runs, no final line break:
    every record decodes to makeUpperCaseAndValidDNA of its sequence, and the split form was really taken."""
    rng = np.random.default_rng(width + 3)
    rs = []
    for i, n in enumerate([700001, 5, 0, 333333, 1, 31, 32, 33, 250000, 64, 1200000, 17]):
        a = U.random_dna(50 + i, max(n, 1))[:n].copy()
        if n > 1000:
            a[n // 3:n // 3 + 777] = ord("N"); a[5] = ord("n"); a[n - 1] = ord("R")
            if width != 1:                                     # '>' inside a line is a base (it becomes N), not a header: only a line's first byte makes one
                for j in (1, 70, 129, n // 2):
                    a[(j // width * width + 1) if width > 1 else j] = ord(">")
            if i % 2:
                a = U.lowercase_some(a, i)
        rs.append(("ctg%d some words" % i, a.tobytes()))
    raw = fasta_bytes(rs, width)
    if width == 60:                                            # blank lines and a carriage return inside a long record (a '\r' is a base: it becomes N)
        cut = raw.index(b"\n", 200000) + 1
        raw = raw[:cut] + b"\n\n" + raw[cut:cut + 30] + b"\r\n" + raw[cut + 30:]
        k = 0
        for j, (h, sq) in enumerate(rs):                       # the same edit in the expectation
            k += len(h) + 2
            nl = (len(sq) + width - 1) // width
            if k + len(sq) + nl > cut:
                o = (cut - k) // (width + 1) * width + 30
                rs[j] = (h, sq[:o] + b"\r" + sq[o:]); break
            k += len(sq) + nl
    raw = raw.rstrip(b"\n")
    path = str(tmp_path / "long.fa")
    open(path, "wb").write(raw)
    want = ["%s\t%d\t%d" % (h.split(" ")[0], len(s), fnv(_normalise(s))) for h, s in rs]
    took_split = 0
    for window, threads in ((1 << 40, 8), (1 << 40, 3), (400000, 5), (65536, 16), (1 << 40, 1)):
        p = subprocess.run([exe, str(window), str(threads), "--packed", path], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout[-300:] + p.stderr
        got = p.stdout.splitlines()
        assert got == want, (width, window, threads, [x for x in zip(got, want) if x[0] != x[1]][:3])
        took_split += int(p.stderr.split("split")[1])
        q = subprocess.run([exe, str(window), str(threads), "--packed", path], capture_output=True, text=True, env=dict(os.environ, MASHMAP_HIP_NO_SPLIT_RECORDS="1"))
        assert q.stdout == p.stdout and int(q.stderr.split("split")[1]) == 0
    assert took_split >= 3
