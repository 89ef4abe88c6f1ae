"""Shared driver for the -m gpu parity tests: run the C-ABI hot path and the CPU oracle on the same
inputs and compare every integer the reference's L1/L2 produce, fragment by fragment."""
import numpy as np

import l1filter
import mmutil as U

FILT_SLOTS, FILT_MAXSPAN, FILT_CONTIGS = 2048, 64, 128      # mm_map.hip: MM_FILT_SLOTS / MM_FILT_MAXSPAN / MM_FILT_CONTIGS


def expected_filtered_points(ix_by_key, ix_points, hashes, min_hits, self_id, seq_counter, flags):
    """what k_filter_points (and k_lookup_mid, same rule) leaves of a fragment's interval points, from the index's per-seed point lists
    (pairs of OPEN, CLOSE in index order, as the device holds them) and tests/l1filter.py's model with the device's hashed bin table;
    sorted as the device sorts them (seqId, pos, CLOSE before OPEN)"""
    seq, o, c = [], [], []
    for hsh in hashes:
        sl = ix_by_key.get(hsh)
        if sl is None:
            continue
        pts = ix_points[sl[0]:sl[1]]
        for j in range(0, len(pts) - 1, 2):
            q = int(pts[j]["seqId"])
            if (flags & U.FLAG_SKIP_SELF) and q == self_id:
                continue
            if (flags & U.FLAG_LOWER_TRI) and not (seq_counter > q):
                continue
            seq.append(q); o.append(int(pts[j]["pos"])); c.append(int(pts[j + 1]["pos"]))
    n = len(seq)
    keep = np.ones(n, dtype=bool)
    gives_up = any(cc <= oo or ((cc - 1) >> l1filter.BIN_SHIFT) - (oo >> l1filter.BIN_SHIFT) >= FILT_MAXSPAN for oo, cc in zip(o, c)) or len(set(seq)) > FILT_CONTIGS
    if n > 1 and min_hits > 1 and not gives_up:
        keep = l1filter.keep_mask(seq, o, c, min_hits, table_slots=FILT_SLOTS)
    out = []
    for i in range(n):
        if keep[i]:
            out.append((seq[i], o[i], 1)); out.append((seq[i], c[i], -1))
    return sorted(out), n - int(keep.sum())


def prefix_groups(names, delim):
    """Map::setRefGroups (computeMap.hpp:144): consecutive contigs sharing name[:rfind(delim)]"""
    pre = [n[:n.rfind(delim)] if delim in n else n for n in names]   # std::string::substr(0, npos) keeps everything
    grp, g = [], -1
    for i, p in enumerate(pre):
        if i == 0 or p != pre[i - 1]:
            g += 1
        grp.append(g)
    return pre, grp


def run_and_compare(oracle, contigs, reads, k=19, L=5000, s=130, pi=0.85, flags=U.FLAG_HG, delim="\0", kmerPct=0.001,
                    seqCounterBase=0, check_points=True, verbose=True, mutate_index=None, device_index=False):
    """contigs: [(name, uint8 array)], reads: [(name, uint8 array)].  Returns (nFragments, nMappedLoci).
    device_index: the context builds its own index from the contigs' bases (mm_index_build: a5-a7 on the device) instead of taking the
    oracle's -- every stage downstream is then checked against the oracle on top of the DEVICE-built index."""
    from mashmap_amd import capi
    h = oracle.session(contigs, k, L, s, pi, U.FILTER_MAP, flags, delim.encode() if delim != "\0" else b"\0", kmerPct, mutate_index=mutate_index)
    ix = oracle.export_index(h)
    cflags = 0
    if flags & U.FLAG_HG: cflags |= capi.MM_FLAG_HG_FILTER
    if flags & U.FLAG_SKIP_SELF: cflags |= capi.MM_FLAG_SKIP_SELF
    if flags & U.FLAG_SKIP_PREFIX: cflags |= capi.MM_FLAG_SKIP_PREFIX
    if flags & U.FLAG_LOWER_TRI: cflags |= capi.MM_FLAG_LOWER_TRIANGULAR
    if flags & U.FLAG_NOSPLIT: cflags |= capi.MM_FLAG_NO_SPLIT        # a read longer than segLength is one fragment (windowLen != 0)
    ctx = capi.Context(k=k, segLength=L, sketchSize=s, flags=cflags)
    cnames = [n for n, _ in contigs]
    refGroup = readGroup = None
    if flags & U.FLAG_SKIP_PREFIX:
        pre, refGroup = prefix_groups(cnames, delim)
        readGroup = []
        for n, _ in reads:
            p = n[:n.rfind(delim)] if delim in n else n
            readGroup.append(refGroup[pre.index(p)] if p in pre else -1)
    selfId = [cnames.index(n) if n in cnames else -1 for n, _ in reads]
    if device_index:
        assert mutate_index is None
        ctx.index_build([a for _, a in contigs], refGroup, kmerPct)
    else:
        ctx.index_upload(ix["minmers"], ix["keys"], ix["offsets"], ix["points"], ix["freq"], ix["contigLen"], refGroup)
    ctx.set_tables(oracle.min_hits_table(s, k, pi), oracle.cutoffs(h))
    ctx.set_replay_tables(*capi.stat_replay_tables(s, k, pi, 0.0, not (flags & U.FLAG_DROP_LOW_ID)))
    nF = ctx.reads_upload([a for _, a in reads], readGroup, selfId, seqCounterBase)
    # first the default path (interval points stay in LDS/registers: fused lookup + sort + L1), then again with the point lists
    # kept in HBM (sort + literal sweep kernels); both must reproduce the reference, and agree with each other
    ctx.map()
    fast = ctx.results()
    ctx.keep_points(True)
    ctx.map()
    stats, l1, l2 = ctx.results()
    for a, b, what in zip(fast, (stats, l1, l2), ("stats", "l1", "l2")):
        assert len(a) == len(b) and a.tobytes() == b.tobytes(), "fused and HBM point paths disagree on " + what
    qsk = ctx.query_sketches()
    frs = ctx.fragments()
    # candidate mappings (k_l2_select: doL2Mapping's best-first walk on the device), fragment-major
    recs = ctx.mappings()
    recs_by_f, f_at = {}, 0
    for m in recs:
        key = (int(m["querySeqId"]) - seqCounterBase, int(m["fragStart"]))
        while (int(frs[f_at]["readId"]), int(frs[f_at]["fragStart"])) != key:
            f_at += 1                                    # records follow the fragment order
        recs_by_f.setdefault(f_at, []).append(m)
    l1_by_f = {}
    for i, c in enumerate(l1):
        l1_by_f.setdefault(int(c["frag"]), []).append((i, c))
    l2_by_c = {}
    for x in l2:
        l2_by_c.setdefault(int(x["cand"]), []).append(x)
    bad = {}
    nloci = 0
    per_frag = {}                                        # fragment -> (post-removal sketch hashes, Q.sketchSize, readId) for the filtered-list check below

    def note(kind, f, got, exp):
        bad[kind] = bad.get(kind, 0) + 1
        if verbose and bad[kind] <= 2:
            print("MISMATCH", kind, "fragment", f, "\n   got", got, "\n   exp", exp)

    for f in range(nF):
        fr = frs[f]
        name, a = reads[int(fr["readId"])]
        seq = a[int(fr["fragStart"]):int(fr["fragStart"]) + int(fr["len"])]
        e = oracle.map_fragment(h, seq, seqCounterBase + int(fr["readId"]), name.encode(), len(a), s)
        st = stats[f]
        if int(st["rawSketchSize"]) != e["rawSketchSize"]: note("rawSketchSize", f, int(st["rawSketchSize"]), e["rawSketchSize"])
        if int(st["sketchSize"]) != e["sketchSize"]: note("sketchSize", f, int(st["sketchSize"]), e["sketchSize"])
        g_sk = [(int(x["hash"]), int(x["strand"])) for x in qsk[f, :int(st["sketchSize"])]]
        e_sk = [(x[0], x[4]) for x in e["sketch"]]
        if g_sk != e_sk: note("sketch", f, g_sk[:4], e_sk[:4])
        per_frag[f] = ([x[0] for x in e_sk], e["sketchSize"], int(fr["readId"]))
        if e["sketchSize"] > 0:
            if check_points:
                gp = [(int(p["seqId"]), int(p["pos"]), int(p["side"])) for p in ctx.points(f)]
                ep = [p[:3] for p in e["points"]]
                if gp != ep: note("points", f, (len(gp), gp[:6]), (len(ep), ep[:6]))
            elif int(st["nPoints"]) != len(e["points"]): note("nPoints", f, int(st["nPoints"]), len(e["points"]))
        g1 = [(int(c["seqId"]), int(c["rangeStartPos"]), int(c["rangeEndPos"]), int(c["intersectionSize"])) for _, c in l1_by_f.get(f, [])]
        if g1 != e["l1"]: note("l1", f, g1[:4], e["l1"][:4])
        else:
            g2 = []
            for ci, (gi, _) in enumerate(l1_by_f.get(f, [])):
                for x in l2_by_c.get(gi, []):
                    g2.append((ci, int(x["seqId"]), int(x["meanOptimalPos"]), int(x["optimalStart"]), int(x["optimalEnd"]),
                               int(x["sharedSketchSize"]), int(x["strand"])))
            nloci += len(g2)
            if g2 != e["l2"]: note("l2", f, g2[:4], e["l2"][:4])
            else:
                # what mapSingleQueryFrag leaves in l2Mappings (computeMap.hpp:774-800), as integers; the oracle's list is sorted by
                # (refSeqId, refStartPos), the device's is in push order: compare as sorted lists
                ql = int(fr["len"])
                gm = sorted((ql, int(m["refStartPos"]), int(m["refStartPos"]) + ql, 0, ql, int(m["refSeqId"]), int(m["querySeqId"]), ql,
                             int(m["sketchSize"]), int(m["conservedSketches"]), int(m["strand"])) for m in recs_by_f.get(f, []))
                em = sorted(x[:11] for x in e["maps_i"])
                if gm != em: note("mappings", f, gm[:3], em[:3])
                for m in recs_by_f.get(f, []):
                    if int(m["rawSketchSize"]) != e["rawSketchSize"] or int(m["fragLen"]) != ql: note("mapping.meta", f, m, e["rawSketchSize"])
    # third run: every list through the HBM path WITH the interval-point pre-filter (MM_OPT_KEEP_POINTS = 2) -- the list the filter leaves
    # must be the model's (tests/l1filter.py with the device's hashed bin table), and L1 / L2 must not notice
    dropped_total = 0
    if check_points and not (flags & (U.FLAG_SKIP_PREFIX | U.FLAG_NOSPLIT)) and not bad:
        ctx.keep_points(2)
        ctx.map()
        for a, b, what in zip(ctx.results(), (stats, l1, l2), ("stats", "l1", "l2")):
            assert len(a) == len(b) and a.tobytes() == b.tobytes(), "the pre-filtered HBM point path disagrees on " + what
        by_key = {int(kk): (int(ix["offsets"][i]), int(ix["offsets"][i + 1])) for i, kk in enumerate(ix["keys"])}
        mh = oracle.min_hits_table(s, k, pi)
        for f in range(nF):
            hashes, qs, rid = per_frag[f]
            if qs <= 0:
                continue
            exp, ndrop = expected_filtered_points(by_key, ix["points"], hashes, int(mh[qs]), selfId[rid], seqCounterBase + rid, flags)
            dropped_total += ndrop
            gp = [(int(p["seqId"]), int(p["pos"]), int(p["side"])) for p in ctx.points(f)]
            if gp != exp: note("filtered points", f, (len(gp), gp[:6]), (len(exp), exp[:6]))
        if verbose:
            print("pre-filter: %d intervals dropped over %d fragments, every filtered list equal to the model's" % (dropped_total, nF))
    ctx.close()
    oracle.free(h)
    assert not bad, "GPU vs oracle mismatches: %r over %d fragments" % (bad, nF)
    return nF, nloci


def fuzz_scenario(seed0, it):
    """a random but reproducible (k, segLength, sketchSize, pi, flags, error rate, genome shape) scenario for run_and_compare"""
    r = U.splitmix64(seed0 * 7919 + it, 16)
    pick = lambda i, xs: xs[int(r[i] % np.uint64(len(xs)))]
    k = pick(0, [15, 16, 17, 19, 19, 19, 21, 24] if seed0 < 3 else [9, 12, 16, 18, 19, 19, 20, 21, 26, 28, 30, 32])
    L = pick(1, [500, 1000, 2000, 5000, 5000, 10000])
    s = pick(2, [10, 20, 40, 64, 65, 128, 130, 130, 200, 310, 498] if seed0 < 3 else [16, 64, 130, 200, 257, 310, 498, 700, 1100, 1279])
    if s > (L - k) // 4: s = max(5, (L - k) // 8)
    pi = pick(3, [0.80, 0.85, 0.85, 0.90, 0.95])
    err = pick(4, [0.0, 0.02, 0.05, 0.10, 0.15])
    flags = pick(5, [U.FLAG_HG, U.FLAG_HG, 0, U.FLAG_HG | U.FLAG_SKIP_SELF, U.FLAG_HG | U.FLAG_LOWER_TRI] +
                 ([U.FLAG_HG | U.FLAG_NOSPLIT, U.FLAG_NOSPLIT, U.FLAG_HG | U.FLAG_NOSPLIT | U.FLAG_SKIP_SELF] if seed0 >= 20 else []))   # campaigns from seed 20 on: --noSplit (windowLen != 0)
    nct = pick(6, [1, 2, 3, 5])
    shape = pick(7, ["random", "random", "repeat", "tandem", "nruns", "dup"])
    kmerPct = pick(8, [0.001, 0.001, 0.0, 0.5])
    sizes = [int(40 * L + (int(r[9 + i % 4]) % (60 * L))) for i in range(nct)]
    cs = []
    for i, n in enumerate(sizes):
        a = U.random_dna(1000 * it + i + seed0 * 100000, n)
        if shape == "repeat":
            unit = a[:max(3 * L, n // 12)]
            a = np.concatenate([U.mutate(unit, 50 + j, 0.01) for j in range(max(2, n // len(unit)))])[:n]
        elif shape == "tandem" and i == 0:
            a = U.tandem_repeat(it, n, int(r[13] % np.uint64(900)) + 5)
        elif shape == "nruns":
            a = U.with_n_runs(a, it, 6, int(L // 3) + 7); a[3] = ord("N")
        cs.append(a)
    if shape == "dup" and nct > 1:
        blk = U.mutate(cs[0][:8 * L], 5, 0.03); cs[1][:len(blk)] = blk[:len(cs[1])][:len(blk)]
    contigs = [("c%d" % i, a) for i, a in enumerate(cs)]
    rl = pick(10, [L, 2 * L, 2 * L + 123, 3 * L + 1, L // 2 + 10])
    reads = [(n_, a) for n_, a, _ in U.sample_reads(cs, it + 3, 40, min(rl, min(len(c) for c in cs)), err)]
    if flags & (U.FLAG_SKIP_SELF | U.FLAG_LOWER_TRI):
        reads = [(contigs[i % nct][0] if i % 3 == 0 else n_, a) for i, (n_, a) in enumerate(reads)]
    desc = dict(it=it, k=k, L=L, s=s, pi=pi, err=err, flags=flags, nct=nct, shape=shape, kmerPct=kmerPct, rl=rl)
    return contigs, reads, dict(k=k, L=L, s=s, pi=pi, flags=flags, kmerPct=kmerPct), desc
