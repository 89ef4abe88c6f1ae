"""The HIP path, through the C ABI, against the golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  No oracle in the loop: this is GPU vs marbl/MashMap v3.1.3 directly."""
import json
import os

import ctypes as C

import numpy as np
import pytest

import mmutil as U
from golden import cases as CS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLD, "golden.json")))


def test_sketch_vs_reference_golden(gold):
    from mashmap_amd import capi
    for name, k, s, seq in CS.sketch_cases():
        ctx = capi.Context(k=k, segLength=max(len(seq), k), sketchSize=s)
        nF = ctx.reads_upload([seq], seqCounterBase=7)
        assert nF == 1
        got, cnt = ctx.sketch()
        g = [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in got[0, :cnt[0]]]
        exp = [(int(h), a, b, c, d) for h, a, b, c, d in gold["sketch"][name]]
        assert g == exp, name
        ctx.close()


def test_index_build_vs_reference_golden(gold):
    """mm_index_build (a5-a7) must reproduce the reference's minmerIndex and lookup-key counts"""
    from mashmap_amd import capi
    contigs, reads, P = CS.session_case()
    z = np.load(os.path.join(GOLD, "session_index.npz"))
    ctx = capi.Context(k=P["k"], segLength=P["segLength"], sketchSize=P["sketchSize"])
    ctx.index_build([a for _, a in contigs], kmerPct=P["kmerPct"])
    ix = ctx.index_download()
    assert len(ix["minmers"]) == gold["session"]["n_minmers"]
    for f in ("hash", "wpos", "wpos_end", "seqId", "strand"):
        assert np.array_equal(ix["minmers"][f], z["minmers"][f]), f
    assert np.array_equal(ix["keys"], z["keys"])
    assert np.array_equal(np.diff(ix["offsets"].astype(np.int64)), z["counts"])
    assert ix["freqThreshold"] == gold["session"]["freq_threshold"]
    ctx.close()


def test_map_vs_reference_golden(gold):
    """index built on the device, tables from the library's own statistics mirror, all L1/L2 integers vs the reference"""
    from mashmap_amd import capi
    contigs, reads, P = CS.session_case()
    ctx = capi.Context(k=P["k"], segLength=P["segLength"], sketchSize=P["sketchSize"], flags=capi.MM_FLAG_HG_FILTER)
    ctx.index_build([a for _, a in contigs], kmerPct=P["kmerPct"])
    ctx.set_tables_default(P["pi"])
    nF = ctx.reads_upload([a for _, a in reads])
    ctx.map()                                            # default path: fused lookup + sort + L1
    fast = ctx.results()
    ctx.keep_points(True)                                # and again with the sorted point lists kept in HBM (for ctx.points)
    ctx.map()
    stats, l1, l2 = ctx.results()
    for a, b in zip(fast, (stats, l1, l2)):
        assert a.tobytes() == b.tobytes()
    frs = ctx.fragments()
    gf = gold["session"]["fragments"]
    assert nF == len(gf)
    l1_by_f, l2_by_c = {}, {}
    for i, c in enumerate(l1):
        l1_by_f.setdefault(int(c["frag"]), []).append((i, c))
    for x in l2:
        l2_by_c.setdefault(int(x["cand"]), []).append(x)
    nl2 = 0
    for f in range(nF):
        e = gf[f]
        assert (int(frs[f]["readId"]), int(frs[f]["fragStart"]), int(frs[f]["len"])) == (e["read"], e["off"], e["len"])
        assert int(stats[f]["sketchSize"]) == e["sketchSize"] and int(stats[f]["rawSketchSize"]) == e["rawSketchSize"]
        gp = [[int(p["seqId"]), int(p["pos"]), int(p["side"])] for p in ctx.points(f)] if e["sketchSize"] else []
        assert gp == e["points"], f
        g1 = [[int(c["seqId"]), int(c["rangeStartPos"]), int(c["rangeEndPos"]), int(c["intersectionSize"])] for _, c in l1_by_f.get(f, [])]
        assert g1 == e["l1"], f
        g2 = []
        for ci, (gi, _) in enumerate(l1_by_f.get(f, [])):
            for x in l2_by_c.get(gi, []):
                g2.append([ci, int(x["seqId"]), int(x["meanOptimalPos"]), int(x["optimalStart"]), int(x["optimalEnd"]),
                           int(x["sharedSketchSize"]), int(x["strand"])])
        assert g2 == e["l2"], f
        nl2 += len(g2)
    assert nl2 > 60
    # the candidate mappings the device selects (k_l2_select) and the floats the product derives from them, against what the real
    # reference's mapSingleQueryFrag reported: integers exactly, identities / upper bound / k-mer complexity to 1e-6
    lib = capi.load()
    recs = ctx.mappings()
    by_frag, at = {}, 0
    for m in recs:
        while (int(frs[at]["readId"]), int(frs[at]["fragStart"])) != (int(m["querySeqId"]), int(m["fragStart"])):
            at += 1
        by_frag.setdefault(at, []).append(m)
    nmap = 0
    for f in range(nF):
        e = gf[f]
        got = []
        for m in by_frag.get(f, []):
            ql, qs, sh = int(m["fragLen"]), int(m["sketchSize"]), int(m["conservedSketches"])
            md = lib.mm_stat_j2md(C.c_float(1.0 * sh / qs), P["k"])
            ident = np.float32(1.0) - np.float32(md)
            ub = np.float32(1.0) - np.float32(lib.mm_stat_md_lower_bound(C.c_float(md), qs, P["k"], C.c_float(0.95)))
            mh01 = float(np.longdouble(int(m["maxHash"])) / np.longdouble(2 ** 64 - 1))
            kc = float(np.float32((float(int(m["rawSketchSize"])) / mh01) / ((ql - P["k"] + 1) * 2)))
            got.append(([ql, int(m["refStartPos"]), int(m["refStartPos"]) + ql, 0, ql, int(m["refSeqId"]), int(m["querySeqId"]), ql, qs, sh, int(m["strand"])],
                        [float(ident), float(ub), kc]))
        got.sort(key=lambda x: (x[0][5], x[0][1]))
        exp = sorted(zip([x[:11] for x in e["maps_i"]], [[float(v) for v in x] for x in e["maps_f"]]), key=lambda x: (x[0][5], x[0][1]))
        assert [g[0] for g in got] == [x[0] for x in exp], f
        for g, x in zip(got, exp):
            assert all(abs(a - b) <= 1e-6 for a, b in zip(g[1], x[1])), (f, g[1], x[1])
        nmap += len(got)
    assert nmap > 50
    ctx.close()


def test_stats_mirror_vs_reference_golden(gold):
    from mashmap_amd import capi
    lib = capi.load()
    st = gold["stats"]
    for i, v in enumerate(st["j2md_130"]):
        assert abs(lib.mm_stat_j2md(i / 130.0, 19) - float(v)) <= 1e-6
    for key, tab in st["min_hits_relaxed"].items():
        s, pi = key.split("_")
        assert [lib.mm_stat_min_hits_relaxed(q, 19, int(pi) / 100.0) for q in range(1, int(s) + 1)] == tab
    assert list(capi.stat_sketch_cutoffs(130, 19, True)) == gold["session"]["cutoffs"]


def test_minmers_vs_reference_golden():
    """device winnowing (k_ref_hash -> candidate compaction -> k_winnow_tiles -> stitch) vs addMinmers of the real reference"""
    from mashmap_amd import capi
    z = np.load(os.path.join(GOLD, "minmers.npz"))
    for name, k, w, s, seq in CS.minmer_cases():
        ctx = capi.Context(k=k, segLength=w, sketchSize=s)
        ctx.index_build([seq], kmerPct=0.0)              # threshold 0 %: no frequent-seed removal, minmerIndex == addMinmers output
        got = ctx.index_download()["minmers"]
        exp = z[name]
        assert len(got) == len(exp), name
        for f in ("hash", "wpos", "wpos_end", "strand"):
            assert np.array_equal(got[f], exp[f]), (name, f)
        ctx.close()
