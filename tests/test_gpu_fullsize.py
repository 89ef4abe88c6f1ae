"""BASELINE.json configs[1] at its full size (1 M x 10 kbp reads = 2 M fragments against a 100 Mbp reference) through the C ABI:

  * the index built on the device equals the oracle's, record for record (5 M minmers, 2.9 M keys, all interval points);
  * the whole batch obeys the invariants the reference's L1/L2 guarantee, and a second pass reproduces it byte for byte;
  * a random sample of reads, mapped again as a small batch against the same index, reproduces its rows of the full batch
    (batch-size independence) and equals the oracle integer for integer (query sketch, L1 candidates, L2 loci),
    so the full-size rows of the sampled reads are the reference's.

Data come from bench.py's generators, so this is the benchmark's own workload."""
import os
import sys

import numpy as np
import pytest

import mmutil as U

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, SEG, SKETCH, PI = 19, 5000, 130, 0.85
N_READS, READ_LEN, N_SAMPLE = 1_000_000, 10_000, 10_000      # 1 % of the reads go through the oracle, integer for integer


def _rows_by_fragment(l1, l2, reads=None, per=2):
    """{fragment: [(seqId, start, end, isize)]}, {fragment: [(candidate rank in fragment, seqId, mean, start, end, shared, strand)]};
    only the fragments of the given (sorted) read ids if any (`per` fragments per read)"""
    i1 = np.arange(len(l1)) if reads is None else np.nonzero(np.isin(l1["frag"] // per, reads))[0]
    i2 = np.arange(len(l2)) if reads is None else np.nonzero(np.isin(l2["frag"] // per, reads))[0]
    first, c1, c2 = {}, {}, {}
    for i in i1:
        c = l1[i]; f = int(c["frag"])
        first.setdefault(f, int(i))
        c1.setdefault(f, []).append((int(c["seqId"]), int(c["rangeStartPos"]), int(c["rangeEndPos"]), int(c["intersectionSize"])))
    for i in i2:
        x = l2[i]; f = int(x["frag"])
        c2.setdefault(f, []).append((int(x["cand"]) - first[f], int(x["seqId"]), int(x["meanOptimalPos"]), int(x["optimalStart"]),
                                     int(x["optimalEnd"]), int(x["sharedSketchSize"]), int(x["strand"])))
    return c1, c2


def test_configs1_full_size(oracle):
    import torch
    sys.path.insert(0, ROOT)
    import bench as B
    from mashmap_amd import capi

    dev = torch.device("cuda", 0)
    W1 = B.WORKLOADS["configs1"]
    contigs_t = B.make_reference(torch, dev, W1["ref_contigs"], W1["ref_contig_len"])
    ref_np = [c.cpu().numpy() for c in contigs_t]
    reads_t = B.make_reads(torch, dev, contigs_t, N_READS, READ_LEN, W1["err"], seed=1000)
    torch.cuda.synchronize()

    # ---- the index, full size, against the oracle ----
    named = [("c%d" % i, a) for i, a in enumerate(ref_np)]
    h = oracle.session(named, K, SEG, SKETCH, PI, U.FILTER_MAP, U.FLAG_HG, b"\0", 0.001)
    e = oracle.export_index(h)
    ctx = capi.Context(k=K, segLength=SEG, sketchSize=SKETCH, flags=capi.MM_FLAG_HG_FILTER)
    ctx.index_build(ref_np, kmerPct=0.001)
    g = ctx.index_download()
    assert len(g["minmers"]) == len(e["minmers"]) > 4_000_000
    for fld in ("hash", "wpos", "wpos_end", "seqId", "strand"):
        assert (g["minmers"][fld] == e["minmers"][fld]).all(), fld
    assert (g["keys"] == e["keys"]).all() and (g["offsets"] == e["offsets"]).all()
    for fld in ("pos", "hash", "seqId", "side"):
        assert (g["points"][fld] == e["points"][fld]).all(), fld
    assert sorted(g["freq"].tolist()) == sorted(e["freq"].tolist())
    del g

    # ---- the whole batch ----
    ctx.set_tables(oracle.min_hits_table(SKETCH, K, PI), oracle.cutoffs(h))
    offs = np.arange(N_READS + 1, dtype=np.int64) * READ_LEN
    nF = ctx.reads_upload_device(reads_t.data_ptr(), reads_t.numel(), offs)
    assert nF == 2 * N_READS
    ctx.map()
    stats, l1, l2 = ctx.results()
    ctx.map()
    stats2, l1b, l2b = ctx.results()
    assert stats.tobytes() == stats2.tobytes() and l1.tobytes() == l1b.tobytes() and l2.tobytes() == l2b.tobytes(), "a second pass differs"
    del stats2, l1b, l2b

    # invariants of sketchSequence / L1 / L2 over all 2 M fragments
    assert (stats["rawSketchSize"] == SKETCH).all()                      # 4 982 k-mers per fragment, no N: the sketch is full
    assert (stats["sketchSize"] <= stats["rawSketchSize"]).all() and (stats["sketchSize"] > 0).all()
    assert (np.diff(l1["frag"].astype(np.int64)) >= 0).all() and (np.diff(l2["frag"].astype(np.int64)) >= 0).all()
    assert int(stats["nL1"].sum()) == len(l1)
    assert (l1["rangeStartPos"] <= l1["rangeEndPos"]).all() and (l1["rangeStartPos"] >= 0).all()
    clen = np.array([len(a) for a in ref_np], dtype=np.int64)
    assert (l1["rangeEndPos"] <= clen[l1["seqId"]]).all()
    minhits = np.asarray(oracle.min_hits_table(SKETCH, K, PI))
    assert (l1["intersectionSize"] >= minhits[stats["sketchSize"][l1["frag"]]]).all()
    cand = l1[l2["cand"]]
    assert (cand["frag"] == l2["frag"]).all() and (cand["seqId"] == l2["seqId"]).all()
    assert (l2["optimalStart"] <= l2["optimalEnd"]).all()
    assert (l2["optimalStart"] >= cand["rangeStartPos"]).all()
    assert (l2["meanOptimalPos"] == (l2["optimalStart"] + l2["optimalEnd"]) // 2).all()
    assert (l2["sharedSketchSize"] <= stats["sketchSize"][l2["frag"]]).all()
    assert (np.abs(l2["strand"]) == 1).all()
    mapped = np.zeros(nF, dtype=bool); mapped[l2["frag"]] = True
    assert mapped.mean() > 0.99                                          # 10 % error reads of a random genome all map

    # ---- a sample of reads: rows of the full batch == small batch == oracle ----
    rng = np.random.default_rng(12345)
    pick = np.sort(rng.choice(N_READS, N_SAMPLE, replace=False))
    idx = torch.from_numpy(pick).to(dev)
    sample = reads_t.view(N_READS, READ_LEN)[idx].cpu().numpy()
    full1, full2 = _rows_by_fragment(l1, l2, pick)
    del reads_t, contigs_t
    nFs = ctx.reads_upload([sample[i] for i in range(N_SAMPLE)])
    assert nFs == 2 * N_SAMPLE
    ctx.map()
    sst, sl1, sl2 = ctx.results()
    qsk = ctx.query_sketches()
    small1, small2 = _rows_by_fragment(sl1, sl2)
    bad = 0
    for i in range(N_SAMPLE):
        for half in range(2):
            fs, ff = 2 * i + half, 2 * int(pick[i]) + half
            assert small1.get(fs, []) == full1.get(ff, []) and small2.get(fs, []) == full2.get(ff, []), "batch-size dependence at read %d" % pick[i]
            assert sst[fs].tobytes() == stats[ff].tobytes()
            seq = sample[i][half * SEG:(half + 1) * SEG]
            ex = oracle.map_fragment(h, seq, i, b"r", READ_LEN, SKETCH)
            g_sk = [(int(x["hash"]), int(x["strand"])) for x in qsk[fs, :int(sst[fs]["sketchSize"])]]
            ok = (int(sst[fs]["rawSketchSize"]) == ex["rawSketchSize"] and g_sk == [(x[0], x[4]) for x in ex["sketch"]]
                  and small1.get(fs, []) == ex["l1"] and small2.get(fs, []) == ex["l2"])
            if not ok:
                bad += 1
                if bad <= 3:
                    print("MISMATCH read", pick[i], "half", half, "\n  got", small1.get(fs), small2.get(fs), "\n  exp", ex["l1"], ex["l2"])
    ctx.close()
    oracle.free(h)
    assert bad == 0, "%d of %d sampled fragments differ from the oracle" % (bad, 2 * N_SAMPLE)


@pytest.mark.parametrize("wl,n_reads,ref_len", [("configs3", 20000, 20_000_000), ("configs4", 12000, 15_000_000)])
def test_other_baseline_shapes_midsize_sampled_parity(oracle, wl, n_reads, ref_len):
    """configs[3] / configs[4] at their read length, error model and sketch size (15 kbp, s = 310; 20 kbp at 15-20 % error, s = 498, pi 80,
    several reference contigs = the --rl list), the reference scaled to what the oracle indexes in seconds: batch invariants on every
    fragment, and >= 1 % of the reads -- re-mapped as a small batch, which must reproduce their rows of the big batch -- equal the oracle
    integer for integer (sketch, Q.sketchSize, L1 candidates, L2 loci)."""
    import torch
    sys.path.insert(0, ROOT)
    import bench as B
    from mashmap_amd import capi
    W = B.WORKLOADS[wl]
    k, seg, s, pi, rl = W["k"], W["seg"], W["sketch"], W["pi"], W["read_len"]
    dev = torch.device("cuda", 0)
    contigs_t = B.make_reference(torch, dev, 4, ref_len // 4)
    ref_np = [c.cpu().numpy() for c in contigs_t]
    reads_t = B.make_reads(torch, dev, contigs_t, n_reads, rl, W["err"], seed=77)
    torch.cuda.synchronize()
    named = [("f%d_c" % i, a) for i, a in enumerate(ref_np)]
    h = oracle.session(named, k, seg, s, pi, U.FILTER_MAP, U.FLAG_HG, b"\0", 0.001)
    ctx = capi.Context(k=k, segLength=seg, sketchSize=s, flags=capi.MM_FLAG_HG_FILTER)
    ctx.index_build(ref_np, kmerPct=0.001)
    g, e = ctx.index_download(), oracle.export_index(h)
    assert len(g["minmers"]) == len(e["minmers"]) and (g["minmers"]["hash"] == e["minmers"]["hash"]).all() and (g["minmers"]["wpos"] == e["minmers"]["wpos"]).all()
    assert (g["keys"] == e["keys"]).all() and (g["offsets"] == e["offsets"]).all()
    del g, e
    ctx.set_tables(oracle.min_hits_table(s, k, pi), oracle.cutoffs(h))
    ctx.set_replay_tables(*capi.stat_replay_tables(s, k, pi))
    offs = np.arange(n_reads + 1, dtype=np.int64) * rl
    nF = ctx.reads_upload_device(reads_t.data_ptr(), reads_t.numel(), offs)
    per = rl // seg
    assert nF == per * n_reads
    ctx.map()
    stats, l1, l2 = ctx.results()
    recs = ctx.mappings()
    assert (stats["rawSketchSize"] == s).all() and (stats["sketchSize"] > 0).all()
    assert int(stats["nL1"].sum()) == len(l1) and (np.diff(l2["frag"].astype(np.int64)) >= 0).all()
    assert (l2["sharedSketchSize"] <= stats["sketchSize"][l2["frag"]]).all()
    mapped = np.zeros(nF, dtype=bool); mapped[l2["frag"]] = True
    assert mapped.mean() > (0.97 if wl == "configs3" else 0.80)
    assert (np.diff(recs["querySeqId"].astype(np.int64)) >= 0).all() and len(recs) >= 0.8 * mapped.sum()
    n_sample = max(200, n_reads // 100)
    rng = np.random.default_rng(4321)
    pick = np.sort(rng.choice(n_reads, n_sample, replace=False))
    sample = reads_t.view(n_reads, rl)[torch.from_numpy(pick).to(dev)].cpu().numpy()
    full1, full2 = _rows_by_fragment(l1, l2, pick, per)
    del reads_t, contigs_t
    assert ctx.reads_upload([sample[i] for i in range(n_sample)]) == per * n_sample
    ctx.map()
    sst, sl1, sl2 = ctx.results()
    qsk = ctx.query_sketches()
    small1, small2 = _rows_by_fragment(sl1, sl2)
    bad = 0
    for i in range(n_sample):
        for j in range(per):
            fs, ff = per * i + j, per * int(pick[i]) + j
            assert small1.get(fs, []) == full1.get(ff, []) and small2.get(fs, []) == full2.get(ff, []), "batch-size dependence at read %d" % pick[i]
            ex = oracle.map_fragment(h, sample[i][j * seg:(j + 1) * seg], i, b"r", rl, s)
            g_sk = [(int(x["hash"]), int(x["strand"])) for x in qsk[fs, :int(sst[fs]["sketchSize"])]]
            if not (int(sst[fs]["rawSketchSize"]) == ex["rawSketchSize"] and g_sk == [(x[0], x[4]) for x in ex["sketch"]]
                    and small1.get(fs, []) == ex["l1"] and small2.get(fs, []) == ex["l2"]):
                bad += 1
    ctx.close(); oracle.free(h)
    assert bad == 0, "%d of %d sampled fragments differ from the oracle" % (bad, per * n_sample)
