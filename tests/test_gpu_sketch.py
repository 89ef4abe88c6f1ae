"""a4 on the device vs the CPU oracle: CommonFunc::sketchSequence (commonFunc.hpp:183).
Bit-exact: hash, first position, last position, strand sign, count -- through the C ABI."""
import numpy as np
import pytest

import mmutil as U

pytestmark = pytest.mark.gpu


def _expect(oracle, reads, k, s, L):
    out = []
    for ri, a in enumerate(reads):
        n = len(a)
        if n < k:
            continue
        if n <= L:
            frs = [a]
        else:
            frs = [a[i * L:(i + 1) * L] for i in range(n // L)]
            if n % L:
                frs.append(a[n - L:])
        for fr in frs:
            out.append(oracle.sketch_sequence(fr, k, s, ri))
    return out


def _check(oracle, reads, k=19, s=130, L=5000):
    from mashmap_amd import capi
    ctx = capi.Context(k=k, segLength=L, sketchSize=s)
    nF = ctx.reads_upload(reads)
    got, cnt = ctx.sketch()
    exp = _expect(oracle, reads, k, s, L)
    assert nF == len(exp)
    bad = 0
    for f in range(nF):
        g = [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in got[f, :cnt[f]]]
        if g != exp[f]:
            bad += 1
            if bad < 3:
                print("fragment", f, "count", cnt[f], len(exp[f]), g[:3], exp[f][:3])
    ctx.close()
    assert bad == 0


def test_sketch_random_reads(oracle):
    g = U.random_dna(1, 400000)
    reads = [a for _, a, _ in U.sample_reads([g], 2, 40, 10000, 0.1)]
    reads += [a for _, a, _ in U.sample_reads([g], 3, 10, 12345, 0.05)]     # overlapping tail fragment
    reads += [U.random_dna(50, 700), U.random_dna(51, 19), U.random_dna(52, 18), U.random_dna(53, 5000), U.random_dna(54, 5001)]
    _check(oracle, reads)


def test_sketch_adversarial(oracle):
    reads = []
    for i in range(12):
        L = 5000 + 777 * i
        a = U.random_dna(100 + i, L)
        if i % 4 == 0: a = U.tandem_repeat(100 + i, L, 23 + 5 * i)          # few distinct k-mers -> hard path
        if i % 4 == 1: a = U.with_n_runs(a, i, 6, 45)
        if i % 4 == 2: a = U.lowercase_some(a, i)
        if i % 4 == 3: a[:] = ord("A")                                      # homopolymer: fwd != rc but one hash
        reads.append(a)
    reads.append(np.frombuffer(b"N" * 6000, dtype=np.uint8).copy())
    reads.append(np.frombuffer(b"ACGT" * 2000, dtype=np.uint8).copy())
    x = U.random_dna(7, 9000); x[100] = ord("N"); x[4990:5010] = ord("n"); x[8999] = ord("X")
    reads.append(x)
    _check(oracle, reads)


# k-mer sizes 1..32 all, 33..64 sampled (the kernels are compiled for every one of them): the word layout of the hash (blocks, 8-byte and shorter tail
# words, the one group shorter than 4 bases) differs for each of them
@pytest.mark.parametrize("k,s,L", [(16, 60, 1000), (19, 498, 5000), (19, 40, 10000), (15, 200, 3000), (21, 130, 5000), (32, 100, 2500), (11, 30, 500),
                                   (17, 70, 1200), (18, 90, 2000), (20, 80, 2000), (22, 64, 1500), (23, 75, 2500), (24, 100, 4000), (25, 50, 1000),
                                   (27, 130, 5000), (29, 40, 800), (31, 100, 3000), (12, 25, 600), (13, 40, 900), (14, 33, 700),
                                   (26, 90, 3000), (28, 120, 4000), (30, 77, 2000),                    # even sizes above 25
                                   (10, 20, 400), (9, 16, 300), (8, 12, 300), (7, 10, 200), (6, 8, 200), (5, 6, 150), (4, 5, 100), (3, 4, 100),
                                   (2, 3, 64), (1, 2, 64),                                             # the reference takes any -k (parseCmdArgs.hpp:435)
                                   (19, 1100, 10000), (19, 1279, 12000),                               # sketches beyond 1024 entries
                                   (19, 1998, 20000),                                                  # --dense --pi 80 -s 20000 (parseCmdArgs.hpp:626-630): hard table spilled to HBM
                                   (19, 4000, 40000), (16, 2500, 15000),                               # beyond the fast kernel's geometry: every fragment on the exact path
                                   (33, 90, 3000), (40, 130, 5000), (47, 60, 2000), (48, 100, 4000),   # k-mers of more than 32 bases: four packed words per strip,
                                   (49, 80, 3000), (56, 120, 5000), (63, 70, 2500), (64, 100, 4000),   # five from 49 on (the reference hashes any length, commonFunc.hpp:138)
                                   (40, 1998, 20000),                                                  # ... on the exact path as well
                                   (19, 9000, 60000), (40, 8500, 50000)])                              # beyond 8 190: the global-memory sketch kernel
def test_sketch_parameter_grid(oracle, k, s, L):
    g = U.random_dna(11, 200000)
    reads = [a for _, a, _ in U.sample_reads([g], 5 + k, 12, 2 * L + 123, 0.08)]
    reads.append(U.with_n_runs(U.random_dna(9, L), 3, 4, 33))
    reads.append(U.tandem_repeat(8, L, 101))
    _check(oracle, reads, k, s, L)


def test_sketch_many_hard_fragments(oracle):
    """a batch in which a large share of the fragments leaves the fast kernel (satellites: 171 distinct k-mers; N runs; tandem repeats):
    the device-resident hard list holds tens of thousands of fragments, every one of them must still come out exact"""
    from mashmap_amd import capi
    rng = np.random.default_rng(77)
    nR, L, k, s = 60000, 5000, 19, 130
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = lut[rng.integers(0, 4, size=(nR, L))]
    kind = rng.integers(0, 4, size=nR)                                       # 0: random, 1: satellite, 2: N run, 3: tandem 2 kbp
    for r in np.nonzero(kind == 1)[0]:
        unit = lut[rng.integers(0, 4, size=171)]
        reads[r] = np.tile(unit, L // 171 + 1)[:L]
    reads[kind == 2, 500:2800] = ord("N")
    for r in np.nonzero(kind == 3)[0][:4000]:
        unit = lut[rng.integers(0, 4, size=50)]
        reads[r, 1000:3000] = np.tile(unit, 40)
    ctx = capi.Context(k=k, segLength=L, sketchSize=s)
    flat = np.ascontiguousarray(reads.reshape(-1))
    nF = ctx.reads_upload([flat[i * L:(i + 1) * L] for i in range(nR)])
    assert nF == nR
    got, cnt = ctx.sketch()
    ctx.close()
    sat = np.nonzero(kind == 1)[0]
    assert len(sat) > 10000 and (cnt[sat] == s).all()                        # 171 distinct canonical hashes >= s
    pick = np.concatenate([rng.choice(np.nonzero(kind == x)[0], 150, replace=False) for x in range(4)])
    bad = 0
    for f in pick:
        e = oracle.sketch_sequence(reads[f], k, s, int(f))
        g = [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in got[f, :cnt[f]]]
        bad += g != e
    assert bad == 0
    # hashes ascending and distinct in every fragment, hard or not
    h = got["hash"].astype(np.uint64)
    inc = (h[:, 1:] > h[:, :-1]) | (np.arange(1, s)[None, :] >= cnt[:, None])
    assert inc.all()


def test_packed_upload_equals_ascii_upload(oracle):
    """mm_reads_upload_packed (host-side normalise + 2-bit pack, pack2bit.hpp) leaves the SAME words in HBM as mm_reads_upload +
    k_pack2bit -- lower case, IUPAC letters, bytes >= 127, N runs, every tail length -- and the sketches that follow are identical"""
    from mashmap_amd import capi
    rng = np.random.default_rng(7)
    reads = [U.random_dna(200 + i, n) for i, n in enumerate((10000, 5000, 5001, 4999, 12345, 31, 32, 33, 19, 18, 1))]
    reads.append(U.lowercase_some(U.random_dna(300, 9000), 3))
    reads.append(U.with_n_runs(U.random_dna(301, 11000), 5, 5, 333))
    reads.append(rng.choice(np.frombuffer(b"ACGTNnacgtRYKMSWBDHV*-.", dtype=np.uint8), 7000))
    weird = U.random_dna(302, 6000); weird[::97] = rng.integers(127, 256, len(weird[::97])).astype(np.uint8); weird[5::211] = 0
    reads.append(weird)
    reads.append(np.zeros(0, dtype=np.uint8))
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130)
    nF = ctx.reads_upload(reads, seqCounterBase=40)
    a2, am, ah = ctx.reads_packed_download()
    sk_a, cnt_a = ctx.sketch()
    for portable in (False, True):
        packed = capi.pack_reads(reads, portable)
        assert a2.tobytes() == packed[0].tobytes() and am.tobytes() == packed[1].tobytes(), "host packing differs from k_pack2bit"
        assert (ah != 0).tolist() == (packed[2] != 0).tolist()
    for prefetch in (False, True):
        assert ctx.reads_upload_packed(packed, seqCounterBase=40, prefetch=prefetch) == nF
        p2, pm, ph = ctx.reads_packed_download()
        assert p2.tobytes() == a2.tobytes() and pm.tobytes() == am.tobytes() and ph.tobytes() == ah.tobytes()
        sk_p, cnt_p = ctx.sketch()
        assert cnt_p.tobytes() == cnt_a.tobytes() and sk_p.tobytes() == sk_a.tobytes()
    # gapped layout (the single-pass parser's threads leave gaps between their pieces): 96 garbage bases behind every read
    groups = (lens.astype(np.int64) + 31) // 32 if False else (packed[3].astype(np.int64) + 31) // 32
    cstart = np.concatenate([[0], np.cumsum(groups)])[:-1] * 32
    gstart = cstart + np.arange(len(groups)) * 96
    total = int(gstart[-1] + groups[-1] * 32 + 96)
    g2 = np.full(total // 16, 0xFFFFFFFF, dtype=np.uint32); gm = np.full(total // 32, 0xFFFFFFFF, dtype=np.uint32)
    for r in range(len(groups)):
        n = int(groups[r])
        g2[gstart[r] // 16:gstart[r] // 16 + 2 * n] = packed[0][cstart[r] // 16:cstart[r] // 16 + 2 * n]
        gm[gstart[r] // 32:gstart[r] // 32 + n] = packed[1][cstart[r] // 32:cstart[r] // 32 + n]
    assert ctx.reads_upload_packed((g2, gm, packed[2], packed[3]), seqCounterBase=40, starts=gstart + 640) == nF
    sk_g, cnt_g = ctx.sketch()
    assert cnt_g.tobytes() == cnt_a.tobytes() and sk_g.tobytes() == sk_a.tobytes(), "gapped packed layout"
    # a sub-block of the batch (what a context of a sharded run gets): words and mask from the block's first read on
    lens = packed[3]
    g0 = int(((lens[:3].astype(np.int64) + 31) // 32).sum())
    sub = (packed[0][2 * g0:], packed[1][g0:], packed[2][3:], lens[3:])
    ctx.reads_upload_packed(sub, seqCounterBase=43)
    sk_s, cnt_s = ctx.sketch()
    f0 = sum(1 for fr in ctx.fragments())
    skip = nF - f0
    assert cnt_s.tobytes() == cnt_a[skip:].tobytes() and sk_s.tobytes() == sk_a[skip:].tobytes()
    ctx.close()
    exp = _expect(oracle, reads, 19, 130, 5000)
    assert nF == len(exp)
    for f in (0, 1, nF - 1):
        g = [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]) - 40, int(x["strand"])) for x in sk_a[f, :cnt_a[f]]]
        assert g == exp[f]


def test_packed_parts_equal_one_upload():
    """mm_reads_upload_packed_parts: several packed pieces (one page-locked reader buffer each in skch::Map) laid end to end as ONE resident
    batch -- the same words in HBM, the same fragments and the same sketches as one upload of the concatenated reads, whether the pieces
    travelled ahead (mm_reads_prefetch_packed_append: all of them, some of them, one more than the upload names) or with the upload;
    pieces with N runs, an empty piece, a gapped piece and reads shorter than k in between"""
    from mashmap_amd import capi
    pieces = [[U.random_dna(400 + i, n) for i, n in enumerate((10000, 4999, 33))],
              [U.with_n_runs(U.random_dna(410, 12000), 4, 5, 200), U.random_dna(411, 18), U.random_dna(412, 7000)],
              [],
              [U.lowercase_some(U.random_dna(420, 15000), 3), U.random_dna(421, 5000)]]
    flat = [r for p in pieces for r in p]
    ctx = capi.Context(k=19, segLength=5000, sketchSize=130)
    nF = ctx.reads_upload(flat, seqCounterBase=7)
    a2, am, ah = ctx.reads_packed_download()
    sk_a, cnt_a = ctx.sketch()
    fr_a = ctx.fragments().tobytes()
    packed = [capi.pack_reads(p) for p in pieces]
    # piece 1 in the gapped layout (64 garbage bases behind every read)
    b2, nm, hasn, lens = packed[1]
    groups = (lens.astype(np.int64) + 31) // 32
    cstart = np.concatenate([[0], np.cumsum(groups)])[:-1] * 32
    gstart = cstart + np.arange(len(groups)) * 64
    total = int(gstart[-1] + groups[-1] * 32)
    g2 = np.full(total // 16, 0xFFFFFFFF, dtype=np.uint32); gm = np.full(total // 32, 0xFFFFFFFF, dtype=np.uint32)
    for r in range(len(groups)):
        n = int(groups[r])
        g2[gstart[r] // 16:gstart[r] // 16 + 2 * n] = b2[cstart[r] // 16:cstart[r] // 16 + 2 * n]
        gm[gstart[r] // 32:gstart[r] // 32 + n] = nm[cstart[r] // 32:cstart[r] // 32 + n]
    parts = [dict(packed=packed[0]), dict(packed=(g2, gm, hasn, lens), starts=gstart + 320), dict(packed=packed[2]), dict(packed=packed[3])]
    extra = capi.pack_reads([U.random_dna(430, 6000)])
    for stage in ((), (0, 1, 2, 3), (1, 3), (3,)):
        if stage == (3,):                                            # one more piece sent ahead than the upload names: it stays staged
            import ctypes as C
            took = C.c_int(-1)
            ctx._ck(ctx.lib.mm_reads_prefetch_packed_append(ctx.h, capi._ptr(extra[0]), capi._ptr(extra[1]), extra[1].size * 32, 1 << 20, C.byref(took)), "append")
            assert took.value == 1                                   # the flag says staged (0: the ring had no room and the piece travels with its upload)
        assert ctx.reads_upload_packed_parts(parts, seqCounterBase=7, stage=stage) == nF
        sk_p, cnt_p = ctx.sketch()
        assert ctx.fragments().tobytes() == fr_a
        assert cnt_p.tobytes() == cnt_a.tobytes() and sk_p.tobytes() == sk_a.tobytes(), stage
    # the piece left staged serves the next upload
    assert ctx.reads_upload_packed_parts([dict(packed=extra)], seqCounterBase=0) == 2          # a 6 000-base read: one full segment + the overlapping tail
    one = capi.Context(k=19, segLength=5000, sketchSize=130)
    one.reads_upload_packed(extra)
    assert one.sketch()[0].tobytes() == ctx.sketch()[0].tobytes()
    one.close()
    # a staged piece whose host words are about to be reused is dropped (mm_reads_prefetch_drop): the upload that names the same two pointers
    # and length afterwards reads the HOST words as they are now, not the stale copy in the staging area
    stale = [x.copy() for x in extra]
    ctx._ck(ctx.lib.mm_reads_prefetch_packed_append(ctx.h, capi._ptr(stale[0]), capi._ptr(stale[1]), stale[1].size * 32, 1 << 20, None), "append")
    ctx._ck(ctx.lib.mm_reads_prefetch_drop(ctx.h), "drop")
    other = capi.pack_reads([U.random_dna(431, 6000)])
    stale[0][:] = other[0]; stale[1][:] = other[1]                   # same buffers, another read's words
    assert ctx.reads_upload_packed_parts([dict(packed=(stale[0], stale[1], other[2], other[3]))], seqCounterBase=0) == 2
    one = capi.Context(k=19, segLength=5000, sketchSize=130)
    one.reads_upload_packed(other)
    assert one.sketch()[0].tobytes() == ctx.sketch()[0].tobytes()
    one.close()
    # without gaps the resident words are those of the single upload, bit for bit
    assert ctx.reads_upload_packed_parts([dict(packed=p) for p in packed], seqCounterBase=7, stage=(0, 3)) == nF
    p2, pm, ph = ctx.reads_packed_download()
    assert p2.tobytes() == a2.tobytes() and pm.tobytes() == am.tobytes() and ph.tobytes() == ah.tobytes()
    ctx.close()
