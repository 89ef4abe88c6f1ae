"""The hot path through the C ABI (sketch -> seed lookup -> L1 sweep -> L2 slide) vs the CPU oracle.
Bit-exact on every integer: Q.sketchSize, the frequent-seed-filtered sketch, the sorted interval points,
L1 candidate regions (computeMap.hpp:916) and L2 loci (computeMap.hpp:1276) for every candidate."""
import numpy as np
import pytest

import mmutil as U
from gpucheck import run_and_compare

pytestmark = pytest.mark.gpu


def genome(seed, sizes, names=None, repeats=True):
    cs = [U.random_dna(seed + i, n) for i, n in enumerate(sizes)]
    if repeats and len(cs) > 1 and min(sizes) >= 150000:
        blk = cs[0][50000:80000]
        mm = U.mutate(blk, 77, 0.03); cs[1][20000:20000 + len(mm)] = mm        # diverged duplicate on another contig
        b2 = cs[0][100000:112000]
        for j in range(4):
            cs[-1][10000 + j * 30000:22000 + j * 30000] = b2                     # 4 exact copies -> repeated seeds
    names = names or ["chr%d" % i for i in range(len(cs))]
    return list(zip(names, cs))


def reads_for(contigs, seed, n, rl, err):
    return [(nm, a) for nm, a, _ in U.sample_reads([c for _, c in contigs], seed, n, rl, err)]


def test_map_default_config(oracle):
    contigs = genome(11, [400000, 300000, 200000])
    reads = reads_for(contigs, 5, 120, 10000, 0.10) + reads_for(contigs, 6, 30, 12345, 0.05)
    reads += [("short", U.random_dna(9, 700)), ("tiny", U.random_dna(10, 18)), ("unrelated", U.random_dna(12, 15000))]
    nF, nl = run_and_compare(oracle, contigs, reads)
    assert nF > 300 and nl > 250


@pytest.mark.parametrize("flags,delim,kmerPct", [(U.FLAG_HG, "\0", 0.001), (U.FLAG_HG, "\0", 0.5), (U.FLAG_HG | U.FLAG_SKIP_PREFIX, "#", 0.001)])
def test_map_against_the_device_built_index(oracle, flags, delim, kmerPct):
    """index -> map stage by stage at small size: mm_index_build (a5-a7 on the device) feeds the map kernels, and every integer of every
    stage must still equal the oracle's -- the other tests upload the ORACLE's index; the full-size test samples"""
    names = ["hapA#1#chr1", "hapA#1#chr2", "hapB#1#chr1"] if delim == "#" else None
    contigs = genome(111, [400000, 300000, 200000], names=names)
    reads = reads_for(contigs, 15, 100, 10000, 0.10) + reads_for(contigs, 16, 20, 7777, 0.04)
    if delim == "#":
        reads = [("hapA#1#r%d" % i if i % 2 else "hapC#9#r%d" % i, a) for i, (_, a) in enumerate(reads)]
    nF, nl = run_and_compare(oracle, contigs, reads, flags=flags, delim=delim, kmerPct=kmerPct, device_index=True)
    assert nF > 200 and nl > 100


@pytest.mark.parametrize("case", ["default", "device_index_freq", "sketch310", "skip_self"])
def test_map_with_the_tagged_seed_table(oracle, monkeypatch, case):
    """the seed table of a human-scale index (buckets of 16 slots behind one tag byte per slot, k_lookup_l1<.., true>) forced onto small
    indexes: same integers as the oracle at every stage; ~0.4 % of the buckets overflow into the next one at this load"""
    monkeypatch.setenv("MM_SEED_TAGS", "1")
    contigs = genome(211, [400000, 300000, 200000])
    reads = reads_for(contigs, 25, 100, 10000, 0.10) + [("unrelated", U.random_dna(12, 15000)), ("short", U.random_dna(9, 700))]
    if case == "default":
        nF, nl = run_and_compare(oracle, contigs, reads)
        assert nF > 200 and nl > 150
    elif case == "device_index_freq":
        run_and_compare(oracle, contigs, reads, kmerPct=0.5, device_index=True)
    elif case == "sketch310":
        run_and_compare(oracle, contigs, reads_for(contigs, 26, 40, 15000, 0.10), s=310)
    else:
        names = [c[0] for c in contigs]
        rs = [(names[i % 3] if i % 4 == 0 else n, a) for i, (n, a) in enumerate(reads)]
        run_and_compare(oracle, contigs, rs, flags=U.FLAG_HG | U.FLAG_SKIP_SELF)


@pytest.mark.parametrize("k", [40, 57])
def test_map_kmers_longer_than_32_bases(oracle, k):
    """-k 40 / 57: every integer of every stage against the oracle (whose getHash is pinned to the reference's at k = 40,
    tests/test_oracle_vs_ref.py); low error so that k-mers of that length still match"""
    contigs = genome(301 + k, [300000, 200000])
    reads = reads_for(contigs, 31, 60, 10000, 0.02) + [("n_runs", U.with_n_runs(contigs[0][1][5000:17000], 3, 6, 40))]
    nF, nl = run_and_compare(oracle, contigs, reads, k=k, device_index=True)
    assert nF > 100 and nl > 50


def test_map_frequent_seeds(oracle):
    contigs = genome(21, [300000, 250000, 200000])
    reads = reads_for(contigs, 7, 80, 10000, 0.08)
    run_and_compare(oracle, contigs, reads, kmerPct=0.5)        # a real frequent-seed set -> Q.sketchSize < s


def test_map_dense_low_identity(oracle):
    contigs = genome(31, [300000, 200000])
    reads = reads_for(contigs, 8, 40, 10000, 0.15)
    run_and_compare(oracle, contigs, reads, s=498, pi=0.80)


def test_map_long_segments_high_identity(oracle):
    contigs = genome(41, [500000, 300000])
    reads = reads_for(contigs, 9, 40, 30000, 0.02)
    run_and_compare(oracle, contigs, reads, L=10000, s=40, pi=0.95)


def test_map_no_hg_filter_small_k(oracle):
    contigs = genome(51, [100000, 80000, 150000], repeats=False)
    reads = reads_for(contigs, 10, 80, 3000, 0.08)
    run_and_compare(oracle, contigs, reads, k=16, L=1000, s=60, flags=0)


def test_map_self_skip_prefix_lower_triangular(oracle):
    # all-vs-all of haplotype-like contigs named S{h}#1#chr{c}: -Y '#' semantics (computeMap.hpp:891-896)
    base = [U.random_dna(60 + c, 60000) for c in range(3)]
    contigs = []
    for hidx in range(3):
        for c in range(3):
            a = base[c] if hidx == 0 else U.mutate(base[c], 100 * hidx + c, 0.02 * hidx)
            contigs.append(("S%d#1#chr%d" % (hidx, c), a))
    reads = [(n, a) for n, a in contigs]
    run_and_compare(oracle, contigs, reads, pi=0.90, flags=U.FLAG_HG | U.FLAG_SKIP_PREFIX, delim="#")
    run_and_compare(oracle, contigs, reads, pi=0.90, flags=U.FLAG_HG | U.FLAG_SKIP_SELF)
    run_and_compare(oracle, contigs, reads, pi=0.90, flags=U.FLAG_HG | U.FLAG_LOWER_TRI)


def test_map_adversarial_reads(oracle):
    contigs = genome(71, [300000, 200000])
    g0 = contigs[0][1]
    reads = [("n_runs", U.with_n_runs(g0[1000:16000], 3, 8, 50)), ("lower", U.lowercase_some(g0[50000:65000], 4)),
             ("tandem", U.tandem_repeat(5, 12000, 31)), ("allN", np.frombuffer(b"N" * 7000, dtype=np.uint8).copy()),
             ("polyA", np.frombuffer(b"A" * 9000, dtype=np.uint8).copy()), ("exact", g0[100000:125000].copy()),
             ("rc", U.revcomp(g0[200000:221000]))]
    run_and_compare(oracle, contigs, reads)


def test_map_repetitive_reference_many_points(oracle):
    # a reference made of a repeated 20 kb unit: every seed has dozens of intervals -> block/global point sorters
    unit = U.random_dna(81, 20000)
    c0 = np.concatenate([U.mutate(unit, 200 + i, 0.01) for i in range(30)])
    contigs = [("rep", c0), ("uniq", U.random_dna(82, 200000))]
    reads = reads_for(contigs, 11, 30, 10000, 0.05)
    run_and_compare(oracle, contigs, reads, kmerPct=0.0)


@pytest.mark.parametrize("seed0,it", [(1, 42), (1, 29), (1, 36), (1, 52)] + [(2, i) for i in range(8)])
def test_map_fuzz_scenarios(oracle, seed0, it):
    """reproducible random scenarios (scripts/fuzz_parity.py runs the long campaign); (1, 42) is a tandem-repeat reference whose
    candidates have more tied L2 loci than the first slot allocation holds"""
    from gpucheck import fuzz_scenario
    contigs, reads, kw, desc = fuzz_scenario(seed0, it)
    run_and_compare(oracle, contigs, reads, verbose=True, **kw)


def test_map_empty_and_degenerate_inputs(oracle):
    """nothing to do must mean nothing done, not a crash: no reads, reads shorter than k, an index without a single minmer"""
    from mashmap_amd import capi
    g = U.random_dna(5, 60000)
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130)
    ctx.index_build([g], kmerPct=0.001)
    ctx.set_tables_default(0.85)
    assert ctx.reads_upload([]) == 0
    ctx.map()
    stats, l1, l2 = ctx.results()
    assert len(stats) == 0 and len(l1) == 0 and len(l2) == 0
    assert ctx.reads_upload([g[:10], g[:18], np.zeros(0, dtype=np.uint8)]) == 0       # all shorter than k: no fragment (computeMap.hpp:325)
    ctx.map()
    assert ctx.result_counts() == (0, 0)
    nF = ctx.reads_upload([g[100:5100], np.frombuffer(b"N" * 6000, dtype=np.uint8).copy()])
    assert nF == 3
    ctx.map()
    stats, l1, l2 = ctx.results()
    assert int(stats[0]["nL1"]) >= 1 and int(stats[1]["sketchSize"]) == 0 and int(stats[2]["sketchSize"]) == 0
    ctx.close()
    # a reference whose contigs are all shorter than segLength has no minmers at all (commonFunc.hpp:341)
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130)
    ctx.index_build([g[:3000], g[3000:7000]], kmerPct=0.001)
    ix = ctx.index_download()
    assert len(ix["minmers"]) == 0 and len(ix["keys"]) == 0
    ctx.set_tables_default(0.85)
    assert ctx.reads_upload([g[:12000]]) == 3
    ctx.map()
    assert ctx.result_counts() == (0, 0)
    ctx.close()


def test_map_short_reads_wide_l2_cells(oracle):
    """reads much shorter than segLength: their sketch spans most of the hash range while a reference window's sketch sits at
    the bottom of it, so over a hundred reference-only hashes pile up below the first query hashes -- more than the 5-bit
    counters of the first L2 pass hold.  Those candidates must be redone with 16-bit cells and still match the reference."""
    from mashmap_amd import capi
    contigs = genome(91, [200000, 150000], repeats=False)
    g0, g1 = contigs[0][1], contigs[1][1]
    reads = [("s%d" % i, g0[10000 + 7000 * i:10000 + 7000 * i + 300 + 40 * i].copy()) for i in range(12)]
    reads += [("t%d" % i, U.revcomp(g1[5000 + 9000 * i:5000 + 9000 * i + 450])) for i in range(8)]
    nF, nl = run_and_compare(oracle, contigs, reads)
    assert nl >= 10
    h = oracle.session(contigs, 19, 5000, 130, 0.85)
    ix = oracle.export_index(h)
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130)
    ctx.index_upload(ix["minmers"], ix["keys"], ix["offsets"], ix["points"], ix["freq"], ix["contigLen"])
    ctx.set_tables_default(0.85)
    ctx.reads_upload([a for _, a in reads])
    ctx.profile(True); ctx.profile_read(reset=True)
    ctx.map()
    launches = ctx.profile_read()["l2"][1]
    assert launches == 2, "the wide-cell pass did not run (launches of the L2 sweep: %d)" % launches
    ctx.close(); oracle.free(h)


# ---- the other BASELINE.json shapes (configs[3], configs[4]) at sizes the oracle finishes in seconds -------------------------------
def test_map_configs3_shape_15kbp_reads_sketch_310(oracle):
    """configs[3]: 15 kbp reads (3 full fragments), 10 % ONT-like error, sketchSize 310 (what the stock binary derives for a 3 GB
    reference file): ~160 interval points per fragment -> the 256-point fused path, LDS tables sized for s = 310"""
    contigs = genome(301, [600000, 500000, 400000])
    reads = reads_for(contigs, 31, 60, 15000, 0.10) + reads_for(contigs, 32, 12, 15000 + 777, 0.10)      # + overlapping tail fragment
    nF, nl = run_and_compare(oracle, contigs, reads, s=310, pi=0.85)
    assert nF >= 3 * 60 + 4 * 12 and nl > 200


def test_map_configs4_shape_20kbp_noisy_reads_dense_pi80_many_files(oracle):
    """configs[4]: --dense --pi 80 (sketchSize 498), 20 kbp reads at 15-20 % error, a reference given as a list of files (--rl): the
    files' contigs share one seqId space (winSketch.hpp:174-214), so at this level they are ten contigs"""
    sizes = [180000, 150000, 120000, 200000, 90000, 110000, 130000, 100000, 160000, 140000]
    contigs = genome(401, sizes, names=["file%d_chr" % i for i in range(10)], repeats=False)
    reads = []
    for i, err in enumerate((0.15, 0.17, 0.20)):
        reads += [("e%d_%s" % (i, n), a) for n, a in reads_for(contigs, 41 + i, 16, 20000, err)]
    nF, nl = run_and_compare(oracle, contigs, reads, s=498, pi=0.80)
    assert nF >= 4 * 48 and nl > 100


def test_map_sketch_beyond_1024(oracle):
    """sketchSize 1100 (> 256 entries per probing batch, > 1024): the multi-batch lookup, seed values through HBM, 16-bit L2 cells"""
    contigs = genome(501, [400000, 300000], repeats=False)
    reads = reads_for(contigs, 51, 20, 20000, 0.12)
    run_and_compare(oracle, contigs, reads, L=10000, s=1100, pi=0.80)


@pytest.mark.parametrize("L,s,pi,err", [(20000, 1998, 0.80, 0.15), (40000, 4000, 0.80, 0.12)])
def test_map_sketch_beyond_1279(oracle, L, s, pi, err):
    """sketch sizes the reference's --dense derives for long segments (0.02 (1 + (1 - pi) / 0.05) (segLength - k), parseCmdArgs.hpp:626-630):
    1998 for --pi 80 -s 20000, 4000 for -s 40000.  Hard sketch table spilled to HBM (beyond ~3400 every fragment takes that path),
    13-bit sketch positions in the L2 stream (s > 2046), 32 / 16 candidates per wave in the sweeps."""
    contigs = genome(551, [500000, 400000], repeats=False)
    reads = reads_for(contigs, 53, 10, 2 * L + 777, err) + reads_for(contigs, 54, 4, L, err / 2)
    run_and_compare(oracle, contigs, reads, L=L, s=s, pi=pi, check_points=False)


def test_map_sketch_beyond_8190(oracle):
    """--dense --pi 80 -s 100000 derives sketchSize 9 998 and the stock binary runs it; here no LDS kernel holds such a sketch, so every
    fragment takes the global-memory sketch kernel (k_sketch_global: all k-mers hashed, sorted, de-duplicated in HBM scratch) and every
    candidate the literal L2 kernel (k_l2_window with windowLen 0).  Every integer of every stage against the oracle, at s = 9 000."""
    L, s = 90000, 9000
    contigs = genome(561, [700000, 500000], repeats=False)
    reads = reads_for(contigs, 55, 6, 2 * L + 1234, 0.12) + reads_for(contigs, 56, 3, L, 0.06) + [("n_runs", U.with_n_runs(contigs[0][1][100000:100000 + L], 3, 5, 60))]
    nF, nl = run_and_compare(oracle, contigs, reads, L=L, s=s, pi=0.80, flags=0, check_points=False)   # (no HG filter: its O(s^3) cut-off table takes the oracle half a minute at this size)
    assert nF >= 20 and nl >= 10


@pytest.mark.parametrize("mode", ["default", "dup_nohg", "prefix"])
def test_map_no_split_reads_longer_than_the_segment(oracle, mode):
    """--noSplit (MM_FLAG_NO_SPLIT): a read longer than segLength is one fragment, windowLen = Q.len - segLength != 0 (computeMap.hpp:933,
    :1309) -- the literal kernels k_l1_window / k_l2_window (hash_to_freq counts, heap of open records in libstdc++ order, shifted
    coordinates) against the oracle, whose windowLen path is pinned to the real reference (test_oracle_vs_ref.py).  Mixed with reads
    shorter than a segment (windowLen == 0 through the same kernels)."""
    cs = [U.random_dna(700 + i, n) for i, n in enumerate((300000, 200000, 150000))]
    flags, delim, names = U.FLAG_HG | U.FLAG_NOSPLIT, "\0", ["chr0", "chr1", "chr2"]
    if mode == "dup_nohg":
        unit = cs[0][40000:52000].copy()
        for j in range(3):
            m = U.mutate(unit, 90 + j, 0.002 * j)[:12000]
            cs[0][100000 + j * 20000:100000 + j * 20000 + len(m)] = m
        cs[1][50000:62000] = unit
        flags = U.FLAG_NOSPLIT
    if mode == "prefix":
        flags |= U.FLAG_SKIP_PREFIX; delim = "#"; names = ["A#1#x", "A#1#y", "B#1#x"]
    contigs = list(zip(names, cs))
    reads = [(nm, a) for nm, a, _ in U.sample_reads(cs, 13, 16, 12345, 0.08)] + [(nm, a) for nm, a, _ in U.sample_reads(cs, 14, 8, 31000, 0.05)]
    reads += [("exact", cs[0][38000:61000].copy()), ("short", cs[1][1000:4000].copy()), ("just_over", cs[1][70000:75001].copy()), ("tiny", cs[2][5:17].copy()),
              ("long80k", U.mutate(cs[2][20000:100000], 5, 0.03))]
    if mode == "prefix":
        reads = [("B#1#r%d" % i if i % 2 else "C#1#r%d" % i, a) for i, (_, a) in enumerate(reads)]
    nF, nl = run_and_compare(oracle, contigs, reads, flags=flags, delim=delim)
    assert nF == len([1 for _, a in reads if len(a) >= 19]) and nl >= 20


def test_map_no_split_read_longer_than_the_lds(oracle):
    """--noSplit with a read that is a whole contig (all-vs-all of assemblies): 450 kbp as ONE fragment does not fit a CU's LDS -- the
    exact sketch kernel then reads the fragment's words from global memory (mm_sketch.hip: stream) -- and windowLen = 445 kbp through
    k_l1_window / k_l2_window.  Every integer against the oracle."""
    g0, g1 = U.random_dna(801, 600000), U.random_dna(802, 300000)
    long_read = U.mutate(g0[100000:550000], 81, 0.03)
    reads = [("contig_as_a_read", long_read), ("r1", U.mutate(g1[20000:31000], 82, 0.05)), ("r2", U.revcomp(U.mutate(g0[5000:9000], 83, 0.02)))]
    nF, nl = run_and_compare(oracle, [("chr0", g0), ("chr1", g1)], reads, flags=U.FLAG_HG | U.FLAG_NOSPLIT, check_points=False)
    assert nF == 3 and nl >= 3


def test_index_with_a_hyper_frequent_seed():
    """a frequent seed's point list is never read on the device (getSeedHits drops the seed first): a list of 2^23 points and more
    (a satellite array in a real genome) must not be refused.  Synthetic index: the resident one plus one such key."""
    from mashmap_amd import capi
    g = U.random_dna(77, 300000)
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130)
    ctx.index_build([g], kmerPct=0.001)
    ix = ctx.index_download()
    ctx.set_tables_default(0.85)
    reads = [a for _, a, _ in U.sample_reads([g], 5, 16, 10000, 0.08)]
    ctx.reads_upload(reads); ctx.map()
    base = ctx.results()
    big = (1 << 23) + 10
    key = np.uint64(0x7fff000000000123)
    assert key not in ix["keys"]
    pts = np.zeros(big, dtype=capi.POINT_DT)
    pts["pos"] = np.arange(big, dtype=np.int64) % 250000; pts["hash"] = key; pts["seqId"] = 0; pts["side"] = np.where(np.arange(big) % 2 == 0, 1, -1)
    keys = np.concatenate([ix["keys"], [key]])
    offs = np.concatenate([ix["offsets"], [ix["offsets"][-1] + big]]).astype(np.uint64)
    points = np.concatenate([ix["points"], pts])
    freq = np.concatenate([ix["freq"], [key]])
    ctx2 = capi.Context(k=19, segLength=5000, sketchSize=130)
    ctx2.index_upload(ix["minmers"], keys, offs, points, freq, np.array([len(g)], dtype=np.int32))
    ctx2.set_tables_default(0.85)
    ctx2.reads_upload(reads); ctx2.map()
    got = ctx2.results()
    for a, b in zip(base, got):
        assert a.tobytes() == b.tobytes()
    ctx.close(); ctx2.close()


def test_map_overlapping_windows_of_one_hash(oracle):
    """An index in which two windows of one hash overlap (addMinmers only removes adjacent duplicates, commonFunc.hpp:560, and a
    loaded index may come from anywhere): a query hash is then open twice during the L2 slide, which the reference's SlideMapper
    absorbs in its own way (boolean `active`, accumulated votes, a count per insert and per delete, slidingMap.hpp:139-145,185-192).
    The fast L2 kernels hand such candidates to k_l2_sweep_exact; every locus must still equal the oracle's."""
    from mashmap_amd import capi
    contigs = genome(601, [300000, 200000], repeats=False)
    reads = reads_for(contigs, 61, 40, 10000, 0.06)

    def overlap(recs):
        extra = []
        for i in range(0, len(recs), 5):
            r = recs[i].copy()
            d = 13 if i % 2 else 37
            r["wpos"] += d; r["wpos_end"] += d
            if i % 3 == 0:
                r["strand"] = -r["strand"]
            extra.append(r)
        allr = np.concatenate([recs, np.array(extra, dtype=recs.dtype)])
        order = np.lexsort((np.arange(len(allr)), allr["wpos"], allr["seqId"]))      # (seqId, wpos), originals first among equals
        return allr[order]

    nF, nl = run_and_compare(oracle, contigs, reads, mutate_index=overlap)
    assert nl > 50
    # the exact kernel really ran
    h = oracle.session(contigs, 19, 5000, 130, 0.85, mutate_index=overlap)
    ix = oracle.export_index(h)
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130)
    ctx.index_upload(ix["minmers"], ix["keys"], ix["offsets"], ix["points"], ix["freq"], ix["contigLen"])
    ctx.set_tables_default(0.85)
    ctx.reads_upload([a for _, a in reads])
    ctx.profile(True); ctx.profile_read(reset=True)
    ctx.map()
    assert ctx.profile_read()["l2"][1] >= 2, "no candidate reached the exact sweep"
    ctx.close(); oracle.free(h)


def test_reads_prefetch_is_invisible(oracle):
    """mm_reads_prefetch: the next batch's bytes copied under the current batch's kernels; the upload that names the same range uses them,
    any other upload ignores them -- results identical either way"""
    from mashmap_amd import capi
    g = U.random_dna(301, 500000)
    contigs = [("c0", g)]
    A = [a for _, a, _ in U.sample_reads([g], 302, 60, 10000, 0.08)]
    B = [a for _, a, _ in U.sample_reads([g], 303, 45, 7777, 0.05)]
    def cat(reads):
        offs = np.zeros(len(reads) + 1, dtype=np.int64); offs[1:] = np.cumsum([len(r) for r in reads])
        return np.ascontiguousarray(np.concatenate(reads)), offs
    bufA, offA = cat(A); bufB, offB = cat(B)
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130, flags=capi.MM_FLAG_HG_FILTER)
    ctx.index_build([g], kmerPct=0.001); ctx.set_tables_default(0.85)
    ctx.reads_upload((bufB, offB)); ctx.map(); wantB = ctx.mappings().tobytes(); wantB2 = ctx.results()[2].tobytes()
    ctx.reads_upload((bufA, offA)); ctx.map(); wantA = ctx.mappings().tobytes()
    for rnd in range(3):
        ctx.reads_upload((bufA, offA))
        ctx.reads_prefetch(bufB)                                            # travels while A is mapped
        ctx.map()
        assert ctx.mappings().tobytes() == wantA
        ctx.reads_upload((bufB, offB)); ctx.map()                           # uses the prefetched bytes
        assert ctx.mappings().tobytes() == wantB and ctx.results()[2].tobytes() == wantB2
    ctx.reads_upload((bufA, offA)); ctx.reads_prefetch(bufB); ctx.map()
    ctx.reads_upload((bufA, offA)); ctx.map()                               # a different range: the prefetch is dropped
    assert ctx.mappings().tobytes() == wantA
    half = bufB[:int(offB[20])]                                          # a view: same address, fewer bytes
    ctx.reads_prefetch(bufB)
    ctx.reads_upload((half, offB[:21].copy())); ctx.map()                   # same start, different length: dropped as well
    ctx.reads_upload((bufB, offB)); ctx.map()
    assert ctx.mappings().tobytes() == wantB
    ctx.close()
    assert len(wantA) > 48 * 100 and len(wantB) > 48 * 40


def test_l2_in_chunks_of_candidates(oracle, monkeypatch):
    """a batch whose located streams exceed MM_L2_STREAM_MIB goes through locate + sweep in chunks of consecutive candidates (repeat-rich
    batches would otherwise size the stream buffer); loci, candidate mappings and their order do not change -- wide and exact re-runs
    included (tandem repeats: tied loci and counter overflows)"""
    from mashmap_amd import capi
    g = U.random_dna(401, 300000)
    rep = U.random_dna(402, 3000)
    for at in (20000, 90000, 150000, 220000):                                 # a dispersed repeat: several candidates per fragment
        m = U.mutate(rep, at, 0.01); n = min(len(m), len(rep)); g[at:at + n] = m[:n]
    g[260000:268000] = U.tandem_repeat(403, 8000, 37)
    reads = [a for _, a, _ in U.sample_reads([g], 404, 80, 6000, 0.05)]
    reads += [g[19000:25000].copy(), g[259000:266000].copy(), U.revcomp(g[89000:95500])]
    def run(budget):
        if budget is None: monkeypatch.delenv("MM_L2_STREAM_MIB", raising=False)
        else: monkeypatch.setenv("MM_L2_STREAM_MIB", budget)
        ctx = capi.Context(k=16, segLength=2000, sketchSize=64, flags=capi.MM_FLAG_HG_FILTER)
        ctx.index_build([g], kmerPct=0.0); ctx.set_tables_default(0.85)
        ctx.reads_upload(reads); ctx.map()
        st, l1, l2 = ctx.results()
        out = (st.tobytes(), l1.tobytes(), l2.tobytes(), ctx.mappings().tobytes(), len(l1), len(l2))
        ctx.close()
        return out
    whole = run(None)
    assert whole[4] > 150 and whole[5] > 150
    for budget in ("0.02", "0.2", "1.5"):
        assert run(budget)[:4] == whole[:4], budget


@pytest.mark.parametrize("limit", ["0", "150"])
def test_l2_exact_kernel_takes_candidates_with_a_long_preload(oracle, monkeypatch, limit):
    """k_l2_locate hands candidates whose pre-load (the records open at rangeStart) is too long for its 12-bit cell counts to
    k_l2_sweep_exact, which replays the pre-load from the index; MM_L2_PRE_LIMIT lowers the limit from 4000 records to where every
    candidate (0) or some of them (150) take that way: same integers as the oracle at every stage"""
    monkeypatch.setenv("MM_L2_PRE_LIMIT", limit)
    contigs = genome(611, [400000, 300000, 200000])
    reads = reads_for(contigs, 35, 100, 10000, 0.10) + reads_for(contigs, 36, 20, 7000, 0.03) + [("unrelated", U.random_dna(12, 15000))]
    nF, nl = run_and_compare(oracle, contigs, reads)
    assert nF > 200 and nl > 150


def test_map_many_candidates_per_fragment(oracle, monkeypatch):
    """a 1.5 kbp unit strewn 90 times over a contig, every copy further than segLength from the next: a read of the unit has ~90 L1
    candidates -- more than k_l1_stream keeps in LDS while counting (its second, writing pass), and the same bytes as the literal kernel"""
    from mashmap_amd import capi
    rng = np.random.default_rng(9)
    unit = U.random_dna(501, 1500)
    g = U.random_dna(502, 90 * 6000 + 4000)
    for i in range(90):
        at = 2000 + i * 6000 + int(rng.integers(0, 800))
        m = U.mutate(unit, 600 + i, 0.004); n = min(len(m), 1500); g[at:at + n] = m[:n]
    contigs = [("rep", g), ("other", U.random_dna(503, 50000))]
    reads = [("u%d" % i, U.mutate(unit, 700 + i, 0.01)[:1400].copy()) for i in range(6)] + [("urc", U.revcomp(unit)), ("bg", g[100:1600].copy())]
    nF, nl = run_and_compare(oracle, contigs, reads, k=16, L=1000, s=80, pi=0.85, kmerPct=0.0)
    assert nF == 2 * len(reads) and nl > 100                                  # 1000 + an overlapping tail fragment per read
    def once(literal):
        if literal: monkeypatch.setenv("MM_L1_LITERAL", "1")
        else: monkeypatch.delenv("MM_L1_LITERAL", raising=False)
        ctx = capi.Context(k=16, segLength=1000, sketchSize=80, flags=capi.MM_FLAG_HG_FILTER)
        ctx.index_build([a for _, a in contigs], kmerPct=0.0); ctx.set_tables_default(0.85)
        ctx.reads_upload([a for _, a in reads]); ctx.map()
        st, l1, l2 = ctx.results()
        ctx.close()
        return st, l1, l2
    a, b = once(False), once(True)
    assert a[0]["nL1"].max() > 64
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes()


def test_steady_state_passes_wait_for_the_device_once(oracle):
    """The first pass of a context sizes every staging buffer from counts it reads back stage by stage; the passes behind it launch
    everything against those capacities with the counts left on the device and wait ONCE (mm_pass_stats).  Same bytes either way
    -- including fragments with more interval points than the fused path holds (a repeated reference: gather + sort + stream kernels
    walk a list whose length stays on the device) -- and a batch that outgrows the buffers is redone the sized way."""
    from mashmap_amd import capi
    unit = U.random_dna(811, 20000)
    rep = np.concatenate([U.mutate(unit, 900 + i, 0.01) for i in range(12)])
    g = U.random_dna(812, 600000)
    contigs = [rep, g]
    small = [a for _, a, _ in U.sample_reads(contigs, 813, 80, 10000, 0.08)]
    big = [a for _, a, _ in U.sample_reads(contigs, 814, 400, 10000, 0.08)]

    def fresh(reads):
        c = capi.Context(k=19, segLength=5000, sketchSize=130, flags=capi.MM_FLAG_HG_FILTER)
        c.index_build(contigs, kmerPct=0.0); c.set_tables_default(0.85)
        c.reads_upload(reads); c.map()
        out = tuple(x.tobytes() for x in c.results()) + (c.mappings().tobytes(),)
        c.close()
        return out

    want_small, want_big = fresh(small), fresh(big)
    assert len(want_small[3]) > 48 * 100
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130, flags=capi.MM_FLAG_HG_FILTER)
    ctx.index_build(contigs, kmerPct=0.0); ctx.set_tables_default(0.85)

    def run(reads):
        ctx.reads_upload(reads); ctx.map()
        return tuple(x.tobytes() for x in ctx.results()) + (ctx.mappings().tobytes(),), ctx.pass_stats()

    got, (syncs, steady) = run(small)
    assert got == want_small and not steady and syncs >= 4            # the sizing pass
    for _ in range(2):
        got, (syncs, steady) = run(small)
        assert got == want_small and steady and syncs == 1, (syncs, steady)
    got, (syncs, steady) = run(big)                                    # five times the batch: the buffers are too small -> redone, sized
    assert got == want_big and not steady and syncs >= 5
    got, (syncs, steady) = run(big)
    assert got == want_big and steady and syncs == 1
    got, (syncs, steady) = run(small)                                  # a smaller batch fits what is there
    assert got == want_small and steady and syncs == 1
    ctx.close()


def test_resident_batches_take_turns_and_an_l1_overflow_in_a_steady_pass_is_redone(oracle):
    """mm_reads_exchange: three uploaded batches stay in HBM and take turns (what bench.py's timed loop does) -- every pass returns the
    bytes a fresh context gives for that batch, the passes behind the first one are steady-state passes as long as the incoming batch
    fits the buffers, and mm_pass_totals counts the one that does not.  The batch that does not is built to overflow the L1 STAGE of a
    steady pass (a read set out of a 30-copy repeat: several times the candidates per fragment the buffers were sized for): the L2 stage
    must not run on what the overflowed L1 stage left (k_l1_gate; the advisor's round-4 finding), the pass is redone and exact."""
    from mashmap_amd import capi
    unit = U.random_dna(821, 20000)
    rep = np.concatenate([U.mutate(unit, 910 + i, 0.01) for i in range(30)])
    g = U.random_dna(822, 600000)
    contigs = [rep, g]
    uniq = lambda seed, n: [a for _, a, _ in U.sample_reads([g], seed, n, 10000, 0.08)]
    A, B = uniq(823, 120), uniq(824, 120)
    # as many fragments as A and B (a batch with a tenth more is sized afresh without a steady attempt), but every one of them has ~30
    # candidate loci and ~2 000 interval points: the sweep path's candidates run past the dense L1 buffer of a pass sized for 240 unique fragments
    C = [a for _, a, _ in U.sample_reads([rep], 825, 120, 10000, 0.05)]

    def fresh(reads):
        c = capi.Context(k=19, segLength=5000, sketchSize=130, flags=capi.MM_FLAG_HG_FILTER)
        c.index_build(contigs, kmerPct=0.0); c.set_tables_default(0.85)
        c.reads_upload(reads); c.map()
        out = tuple(x.tobytes() for x in c.results()) + (c.mappings().tobytes(),)
        n1 = c.result_counts()[0]
        c.close()
        return out, n1

    (wantA, nA), (wantB, nB), (wantC, nC) = fresh(A), fresh(B), fresh(C)
    assert nC > 4 * nA                                                           # the repeat batch really outgrows the candidate buffers
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130, flags=capi.MM_FLAG_HG_FILTER)
    ctx.index_build(contigs, kmerPct=0.0); ctx.set_tables_default(0.85)
    nFA = ctx.reads_upload(A); ctx.reads_exchange(0)                             # slot 0 = A
    ctx.reads_upload(C); ctx.reads_exchange(1)                                   # slot 1 = C
    nFB = ctx.reads_upload(B)                                                    # resident = B
    assert ctx.num_fragments() == nFB > 200
    got = lambda: tuple(x.tobytes() for x in ctx.results()) + (ctx.mappings().tobytes(),)
    ctx.map(); assert got() == wantB and not ctx.pass_stats()[1]                 # sizing pass
    ctx.reads_exchange(0); ctx.map()                                             # resident A, slot 0 = B
    assert got() == wantA and ctx.pass_stats() == (1, True)
    ctx.reads_exchange(0); ctx.map()                                             # resident B again
    assert got() == wantB and ctx.pass_stats() == (1, True)
    t = ctx.pass_totals(); assert t == {"passes": 3, "steady": 2, "redone": 0}, t
    ctx.reads_exchange(1); ctx.map()                                             # resident C: L1 candidates overflow the steady pass
    assert got() == wantC and not ctx.pass_stats()[1]
    t = ctx.pass_totals(); assert t == {"passes": 4, "steady": 2, "redone": 1}, t
    ctx.map(); assert got() == wantC and ctx.pass_stats() == (1, True)           # sized for C now
    ctx.reads_exchange(0); ctx.map(); assert got() == wantA and ctx.pass_stats() == (1, True)           # slot 0 held A (B sits in slot 1 since C came out of it)
    ctx.reads_exchange(2); assert ctx.num_fragments() == 0                       # an empty slot: nothing resident
    ctx.reads_exchange(2); assert ctx.num_fragments() == nFA
    with pytest.raises(capi.MashmapError):
        ctx.reads_exchange(capi.load().mm_abi_version() + 99)
    ctx.close()


def test_pass_counts_report_the_hard_list(oracle):
    """mm_pass_stats' fifth count: fragments the fast sketch kernel handed to the exact one -- none for random reads, every fragment that is
    nothing but a short tandem repeat (fewer than sketchSize distinct k-mers)"""
    from mashmap_amd import capi
    g = U.random_dna(831, 300000)
    reads = [a for _, a, _ in U.sample_reads([g], 832, 20, 10000, 0.05)]
    sat = np.tile(U.random_dna(833, 171), 10000 // 171 + 1)[:10000]
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130, flags=capi.MM_FLAG_HG_FILTER)
    ctx.index_build([g], kmerPct=0.0); ctx.set_tables_default(0.85)
    ctx.reads_upload(reads); ctx.map()
    assert ctx.pass_counts()["hard"] == 0
    ctx.reads_upload(reads + [sat, sat]); ctx.map()
    assert ctx.pass_counts()["hard"] == 4                                        # two fragments per satellite read
    ctx.map()
    assert ctx.pass_stats()[1] and ctx.pass_counts()["hard"] == 4                # the count comes back with a steady-state pass's counters too
    ctx.close()


def test_reserved_fragments_make_a_growing_batch_a_steady_pass(oracle):
    """MM_OPT_RESERVE_FRAGMENTS: the pass that sizes the staging buffers sizes them for the announced batch, so a batch five times the first
    one -- what skch::Map's device passes do when they grow from one reader batch to four -- goes through as a steady-state pass (one host
    wait, no reallocation) with the bytes of a fresh context; without the announcement it is sized again (the test above)"""
    from mashmap_amd import capi
    g = U.random_dna(842, 600000)
    contigs = [g]                                                                # (one kind of read: the counts of a batch scale with its size, as the 51 k-read batches of a real run do)
    small = [a for _, a, _ in U.sample_reads(contigs, 843, 200, 10000, 0.08)]
    big = [a for _, a, _ in U.sample_reads(contigs, 844, 1000, 10000, 0.08)]

    def fresh(reads):
        c = capi.Context(k=19, segLength=5000, sketchSize=130, flags=capi.MM_FLAG_HG_FILTER)
        c.index_build(contigs, kmerPct=0.0); c.set_tables_default(0.85)
        n = c.reads_upload(reads); c.map()
        out = tuple(x.tobytes() for x in c.results()) + (c.mappings().tobytes(),)
        c.close()
        return out, n

    (want_small, _), (want_big, nBig) = fresh(small), fresh(big)
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130, flags=capi.MM_FLAG_HG_FILTER)
    ctx.index_build(contigs, kmerPct=0.0); ctx.set_tables_default(0.85)
    ctx.reserve_fragments(nBig + nBig // 8)
    got = lambda: tuple(x.tobytes() for x in ctx.results()) + (ctx.mappings().tobytes(),)
    ctx.reads_upload(small); ctx.map()
    assert got() == want_small and not ctx.pass_stats()[1]
    ctx.reads_upload(big); ctx.map()
    assert got() == want_big and ctx.pass_stats() == (1, True), ctx.pass_stats()
    ctx.reads_upload(small); ctx.map()
    assert got() == want_small and ctx.pass_stats() == (1, True)
    assert ctx.pass_totals() == {"passes": 3, "steady": 2, "redone": 0}
    ctx.close()
