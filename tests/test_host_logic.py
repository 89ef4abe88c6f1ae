"""The device-independent half of the host side (mashmap_amd/host/skch_map_post.hpp: best-first doL2Mapping replay, chaining,
plane-sweep filters, boundary checks) on the CPU: it is fed the integers the kernels would produce -- here taken from the real
reference's own L1/L2 stages, or from the committed golden vectors -- and must reproduce what the reference's mapModule
(computeMap.hpp:570-714) returns for the read."""
import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import mmutil as U
from golden import cases as CS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HL_DIR = os.path.join(ROOT, "tests", "hostlogic")
GOLD = os.path.join(ROOT, "tests", "golden")

FRAG_DT = np.dtype([("readId", "<i4"), ("fragStart", "<i4"), ("len", "<i4"), ("pad", "<i4")])
STATS_DT = np.dtype([("rawSketchSize", "<i4"), ("sketchSize", "<i4"), ("maxHash", "<u8"), ("nPoints", "<i4"), ("nL1", "<i4")])
L1_DT = np.dtype([("frag", "<i4"), ("seqId", "<i4"), ("rangeStartPos", "<i4"), ("rangeEndPos", "<i4"), ("intersectionSize", "<i4")])
L2_DT = np.dtype([("frag", "<i4"), ("cand", "<i4"), ("seqId", "<i4"), ("meanOptimalPos", "<i4"), ("optimalStart", "<i4"),
                  ("optimalEnd", "<i4"), ("sharedSketchSize", "<i4"), ("strand", "<i4")])


@pytest.fixture(scope="module")
def hl():
    so = os.path.join(HL_DIR, "libhostlogic.so")
    src = os.path.join(HL_DIR, "hostlogic.cpp")
    hdrs = [os.path.join(ROOT, "mashmap_amd", "host", f) for f in ("skch_map_post.hpp", "skch_types.hpp", "mm_stats.hpp")]
    hdrs += [os.path.join(ROOT, "mashmap_amd", "csrc", f) for f in ("mm_select_core.h", "mm_heap.h")]
    if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in [src] + hdrs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wno-sign-compare", "-o", so, src])
    lib = C.CDLL(so)
    lib.hl_map_read.restype = C.c_int
    return lib


def run_host_logic(lib, P, contigs, groups, name, read_len, seq_counter, frag_rows, flags=U.FLAG_HG, filter_mode=U.FILTER_MAP, n_mappings=1):
    """frag_rows: list of dicts(off, len, sketchSize, rawSketchSize, maxHash, l1, l2) in fragment order"""
    nF = len(frag_rows)
    frags = np.zeros(nF, dtype=FRAG_DT); stats = np.zeros(nF, dtype=STATS_DT)
    l1rows, l2rows = [], []
    for f, fr in enumerate(frag_rows):
        frags[f] = (0, fr["off"], fr["len"], 0)
        stats[f] = (fr["rawSketchSize"], fr["sketchSize"], fr["maxHash"], 0, len(fr["l1"]))
        base = len(l1rows)
        for c in fr["l1"]:
            l1rows.append((f,) + tuple(c))
        for x in fr["l2"]:
            l2rows.append((f, base + x[0]) + tuple(x[1:]))
    l1 = np.array(l1rows, dtype=L1_DT) if l1rows else np.zeros(0, dtype=L1_DT)
    l2 = np.array(l2rows, dtype=L2_DT) if l2rows else np.zeros(0, dtype=L2_DT)
    names = (C.c_char_p * len(contigs))(*[n.encode() for n, _ in contigs])
    lens = np.array([len(a) for _, a in contigs], dtype=np.int32)
    grp = np.array(groups, dtype=np.int32) if groups is not None else None
    out = (U.Mapping * 256)(); paf = C.create_string_buffer(1 << 16)
    vp = C.c_void_p
    n = lib.hl_map_read(C.c_int(P["k"]), C.c_int(P["segLength"]), C.c_int(P["sketchSize"]), C.c_float(P["pi"]), C.c_int(filter_mode),
                        C.c_int(flags), C.c_int(n_mappings), C.c_int(len(contigs)), names, lens.ctypes.data_as(vp),
                        grp.ctypes.data_as(vp) if grp is not None else None, name.encode(), C.c_int(read_len), C.c_int(seq_counter),
                        C.c_int(nF), frags.ctypes.data_as(vp), stats.ctypes.data_as(vp), C.c_int(len(l1)), l1.ctypes.data_as(vp),
                        C.c_int(len(l2)), l2.ctypes.data_as(vp), out, C.c_int(256), paf, C.c_int(1 << 16))
    assert n != -2, "the integer walk of k_l2_select + MapPost::mapModuleFromRecords disagree with the float replay"
    assert 0 <= n <= 256
    return [out[i].ikey() for i in range(n)], [out[i].fkey() for i in range(n)], paf.value.decode()


def _ref_map_read(ref, h, a, ri, name):
    buf = (U.Mapping * 256)()
    n = ref.f("session_map_read")(h, bytes(a), len(a), ri, name.encode(), buf, 256)
    return [buf[i].ikey() for i in range(n)], [buf[i].fkey() for i in range(n)]


@pytest.mark.parametrize("mode", ["default", "pi90_n2", "nomerge", "filter_none"])
def test_host_logic_vs_reference_mapmodule(hl, ref, mode):
    cs = [U.random_dna(800 + i, n) for i, n in enumerate((260000, 200000, 90000))]
    blk = U.mutate(cs[0][30000:70000], 9, 0.03); cs[1][10000:10000 + len(blk)] = blk          # competing locus on another contig
    contigs = list(zip(["chr0", "chr1", "chr2"], cs))
    reads = [(nm, a) for nm, a, _ in U.sample_reads(cs, 3, 30, 10000, 0.10)]
    reads += [(nm + "l", a) for nm, a, _ in U.sample_reads(cs, 4, 10, 23456, 0.05)]
    reads += [("dupread", cs[0][32000:52000].copy()), ("chim", np.concatenate([cs[0][100000:112000], U.revcomp(cs[2][20000:31000])]))]
    P = dict(k=19, segLength=5000, sketchSize=130, pi=0.85)
    flags, fmode, nmap = U.FLAG_HG, U.FILTER_MAP, 1
    if mode == "pi90_n2": P["pi"] = 0.90; nmap = 2
    if mode == "nomerge": flags |= U.FLAG_NOMERGE
    if mode == "filter_none": flags = 0; fmode = U.FILTER_NONE
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "r.fa")
        U.write_fasta(fa, contigs)
        h = ref.session([fa], P["k"], P["segLength"], P["sketchSize"], P["pi"], fmode, flags, b"\0", 0.001, nmap)
        nmapped = 0
        for ri, (nm, a) in enumerate(reads):
            rows = []
            for off, ln in CS.fragments_of(len(a), P["segLength"]):
                seq = a[off:off + ln]
                e = ref.map_fragment(h, seq, ri, nm.encode(), len(a), P["sketchSize"])
                raw = ref.sketch_sequence(seq, P["k"], P["sketchSize"], ri)
                rows.append(dict(off=off, len=ln, sketchSize=e["sketchSize"], rawSketchSize=e["rawSketchSize"],
                                 maxHash=raw[-1][0] if raw else 0, l1=e["l1"], l2=e["l2"]))
            gi, gf, paf = run_host_logic(hl, P, contigs, None, nm, len(a), ri, rows, flags, fmode, nmap)
            ei, ef = _ref_map_read(ref, h, a, ri, nm)
            assert gi == ei, (mode, nm, gi[:3], ei[:3])
            for x, y in zip(gf, ef):
                assert all(abs(p - q) <= 1e-6 for p, q in zip(x, y))
            assert paf.count("\n") == len(ei)
            nmapped += len(ei)
        ref.free(h)
    assert nmapped > 30


def test_host_logic_paf_text_vs_golden(hl):
    """without the reference: per-fragment integers from tests/golden/golden.json (recorded from the real reference), PAF lines
    of the same reads from tests/golden/paf -- not available for this session case, so the check here is structural: every
    mapping the golden fragments imply comes out once, inside the read and the contig"""
    gold = json.load(open(os.path.join(GOLD, "golden.json")))
    contigs, reads, P = CS.session_case()
    frs = gold["session"]["fragments"]
    by_read = {}
    for fr in frs:
        by_read.setdefault(fr["read"], []).append(fr)
    total = 0
    for ri, (nm, a) in enumerate(reads):
        rows = []
        for fr in by_read.get(ri, []):
            # golden keeps the post-removal sketch head only; kmerComplexity needs the raw sketch's largest hash: recover it from
            # the recorded complexity is not possible, so take it from the oracle-independent formula input stored with the fragment
            rows.append(dict(off=fr["off"], len=fr["len"], sketchSize=fr["sketchSize"], rawSketchSize=fr["rawSketchSize"],
                             maxHash=int(fr["sketch_last"]) if fr["sketch_last"] else 0, l1=[tuple(x) for x in fr["l1"]],
                             l2=[tuple(x) for x in fr["l2"]]))
        if not rows:
            continue
        gi, gf, paf = run_host_logic(hl, P | {"pi": P["pi"]}, contigs, None, nm, len(a), ri, rows)
        for m in gi:
            qlen, rs, re_, qs, qe, rid = m[0], m[1], m[2], m[3], m[4], m[5]
            assert qlen == len(a) and 0 <= qs <= qe <= len(a) and 0 <= rs <= re_ < len(contigs[rid][1])
        lines = [l for l in paf.splitlines() if l]
        assert len(lines) == len(gi) and all(l.split("\t")[0] == nm and len(l.split("\t")) >= 12 for l in lines)
        total += len(gi)
    assert total >= 30


def test_paf_text_by_to_chars_equals_the_stream_form(hl):
    """MapPost::appendReadMappings (field by field with std::to_chars: what the post stage runs) against reportReadMappingsStream (the
    reference's insertions into a stream, computeMap.hpp:1758-1806) on random mappings -- identity 0, 1, three-digit and eight-digit
    values, long double complexities -- in all 16 combinations of legacy / percentage / no-merge / one-to-one output"""
    hl.hl_paf_formatters_agree.restype = C.c_int
    hl.hl_paf_formatters_agree.argtypes = [C.c_int, C.c_ulonglong]
    for seed in (1, 2, 3):
        assert hl.hl_paf_formatters_agree(20000, seed) == 0


@pytest.mark.parametrize("s,k,pi,keep_low", [(130, 19, 0.85, 1), (310, 19, 0.85, 1), (498, 16, 0.80, 1), (700, 19, 0.95, 1), (310, 40, 0.90, 0)])
def test_integer_tables_equal_their_literal_forms(hl, s, k, pi, keep_low):
    """replayTables / minHitsTable (mm_stats.hpp) against the literal loops: the binomial tail summed over all of its terms, the L1 cut-off
    counted up as computeMap.hpp:1196-1200 does -- the shortened forms (tail sum stopped where a term can no longer change the sum, cut-off
    bisected) must give every entry the same value"""
    hl.hl_tables_vs_literal.restype = C.c_int
    assert hl.hl_tables_vs_literal(C.c_int(s), C.c_int(k), C.c_float(pi), C.c_int(keep_low)) == 0


def test_binomial_tail_equals_the_full_sum_bit_for_bit(hl):
    hl.hl_tail_vs_full.restype = C.c_int
    assert hl.hl_tail_vs_full() == 0
