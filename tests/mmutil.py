"""Shared helpers for the test-suite, bench.py's cpu_baseline leg and __graft_entry__.smoke().

* deterministic synthetic DNA (counter-based splitmix64, so fixtures regenerate bit-identically
  on any numpy version),
* ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when it has been built from
  /root/reference, the real reference (oracle/_ref/libmashmap_ref.so).

Nothing in here is imported by the product package (mashmap_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libmashmap_ref.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "mashmap_ref")

# ----------------------------------------------------------------------------- synthetic data
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, offset=0):
    """n pseudo-random uint64 values: splitmix64 of (seed, counter)."""
    with np.errstate(over="ignore"):
        x = (np.arange(offset, offset + n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        x = x + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[a] = b


def random_dna(seed, n):
    """uniform random ACGT, as a uint8 numpy array of ASCII codes"""
    return _ACGT[(splitmix64(seed, n) >> np.uint64(62)).astype(np.int64)]


def revcomp(a):
    return _COMP[a[::-1]]


def mutate(a, seed, err):
    """ONT-like i.i.d. errors: err/3 substitutions, err/3 insertions, err/3 deletions."""
    n = len(a)
    r = splitmix64(seed, n)
    u = (r >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    kind = np.zeros(n, dtype=np.int8)          # 0 keep, 1 sub, 2 ins (before base), 3 del
    kind[u < err] = 3
    kind[u < 2 * err / 3] = 2
    kind[u < err / 3] = 1
    newb = _ACGT[((r >> np.uint64(3)) & np.uint64(3)).astype(np.int64)]
    sub = a.copy()
    m = kind == 1
    # substitution: rotate to a different base
    idx = np.searchsorted(_ACGT, a[m]) if m.any() else np.zeros(0, dtype=np.int64)
    sub[m] = _ACGT[(idx + 1 + ((r[m] >> np.uint64(5)) % np.uint64(3)).astype(np.int64)) % 4]
    keep = kind != 3
    counts = keep.astype(np.int64) + (kind == 2)
    out = np.empty(int(counts.sum()), dtype=np.uint8)
    pos = np.cumsum(counts) - counts
    ins = kind == 2
    out[pos[ins]] = newb[ins]
    out[pos[keep] + ins[keep]] = sub[keep]
    return out


def tandem_repeat(seed, n, unit=37):
    u = random_dna(seed, unit)
    return np.tile(u, n // unit + 1)[:n].copy()


def with_n_runs(a, seed, nruns=5, runlen=40):
    a = a.copy()
    starts = (splitmix64(seed, nruns) % np.uint64(max(1, len(a) - runlen))).astype(np.int64)
    for s in starts:
        a[s:s + runlen] = ord("N")
    return a


def lowercase_some(a, seed, frac=0.3):
    a = a.copy()
    r = splitmix64(seed, len(a))
    m = (r >> np.uint64(40)).astype(np.float64) / float(1 << 24) < frac
    a[m] = a[m] + 32
    return a


def sample_reads(genome_contigs, seed, nreads, readlen, err):
    """reads sampled uniformly from the contigs, random strand, mutated.  Returns list of (name, uint8 array, truth)."""
    lens = np.array([len(c) for c in genome_contigs], dtype=np.int64)
    ok = lens >= readlen
    r = splitmix64(seed, 3 * nreads)
    out = []
    cand = np.nonzero(ok)[0]
    for i in range(nreads):
        ci = int(cand[int(r[3 * i] % np.uint64(len(cand)))])
        st = int(r[3 * i + 1] % np.uint64(lens[ci] - readlen + 1))
        strand = int(r[3 * i + 2] & np.uint64(1))
        frag = genome_contigs[ci][st:st + readlen]
        if strand:
            frag = revcomp(frag)
        if err > 0:
            frag = mutate(frag, seed * 1000003 + i, err)
        out.append(("read%d" % i, np.ascontiguousarray(frag), (ci, st, strand)))
    return out


def write_fasta(path, records, width=80):
    with open(path, "wb") as f:
        for name, a in records:
            f.write(b">" + name.encode() + b"\n")
            b = a.tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")


# ----------------------------------------------------------------------------- C structs
class Minmer(C.Structure):
    _fields_ = [("hash", C.c_uint64), ("wpos", C.c_int32), ("wpos_end", C.c_int32), ("seqId", C.c_int32),
                ("strand", C.c_int16), ("pad", C.c_int16)]

    def key(self):
        return (self.hash, self.wpos, self.wpos_end, self.seqId, self.strand)


class Point(C.Structure):
    _fields_ = [("pos", C.c_int32), ("pad0", C.c_int32), ("hash", C.c_uint64), ("seqId", C.c_int32),
                ("side", C.c_int8), ("pad1", C.c_int8 * 3)]

    def key(self):
        return (self.seqId, self.pos, self.side, self.hash)


class L1(C.Structure):
    _fields_ = [("seqId", C.c_int32), ("rangeStartPos", C.c_int32), ("rangeEndPos", C.c_int32),
                ("intersectionSize", C.c_int32)]

    def key(self):
        return (self.seqId, self.rangeStartPos, self.rangeEndPos, self.intersectionSize)


class L2(C.Structure):
    _fields_ = [("seqId", C.c_int32), ("meanOptimalPos", C.c_int32), ("optimalStart", C.c_int32),
                ("optimalEnd", C.c_int32), ("sharedSketchSize", C.c_int32), ("strand", C.c_int32)]

    def key(self):
        return (self.seqId, self.meanOptimalPos, self.optimalStart, self.optimalEnd, self.sharedSketchSize, self.strand)


class Mapping(C.Structure):
    _fields_ = [("queryLen", C.c_int32), ("refStartPos", C.c_int32), ("refEndPos", C.c_int32),
                ("queryStartPos", C.c_int32), ("queryEndPos", C.c_int32), ("refSeqId", C.c_int32),
                ("querySeqId", C.c_int32), ("blockLength", C.c_int32), ("nucIdentity", C.c_float),
                ("nucIdentityUpperBound", C.c_float), ("sketchSize", C.c_int32), ("conservedSketches", C.c_int32),
                ("strand", C.c_int32), ("approxMatches", C.c_int32), ("kmerComplexity", C.c_double)]

    def ikey(self):
        return (self.queryLen, self.refStartPos, self.refEndPos, self.queryStartPos, self.queryEndPos, self.refSeqId,
                self.querySeqId, self.blockLength, self.sketchSize, self.conservedSketches, self.strand,
                self.approxMatches)

    def fkey(self):
        return (self.nucIdentity, self.nucIdentityUpperBound, self.kmerComplexity)


FLAG_HG, FLAG_SKIP_SELF, FLAG_SKIP_PREFIX, FLAG_LOWER_TRI, FLAG_NOSPLIT, FLAG_NOMERGE, FLAG_DROP_LOW_ID = \
    1, 2, 4, 8, 16, 32, 64
FILTER_MAP, FILTER_ONETOONE, FILTER_NONE = 1, 2, 3


def build_oracle():
    """compile oracle/liboracle.so (and oracle/_ref when /root/reference is present)"""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


def _bind_common(lib, p):
    g = lambda n: getattr(lib, p + n)
    g("get_hash").restype = C.c_uint64
    g("get_hash").argtypes = [C.c_char_p, C.c_int]
    g("sketch_sequence").restype = C.c_int
    g("sketch_sequence").argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Minmer), C.c_int]
    g("add_minmers").restype = C.c_int64
    g("add_minmers").argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Minmer), C.c_int64]
    g("j2md").restype = C.c_float; g("j2md").argtypes = [C.c_float, C.c_int]
    g("md2j").restype = C.c_float; g("md2j").argtypes = [C.c_float, C.c_int]
    g("md_lower_bound").restype = C.c_float; g("md_lower_bound").argtypes = [C.c_float, C.c_int, C.c_int, C.c_float]
    g("min_hits").restype = C.c_int; g("min_hits").argtypes = [C.c_int, C.c_int, C.c_float]
    g("min_hits_relaxed").restype = C.c_int; g("min_hits_relaxed").argtypes = [C.c_int, C.c_int, C.c_float]
    g("recommended_sketch_size").restype = C.c_int64
    g("recommended_sketch_size").argtypes = [C.c_int, C.c_float, C.c_int64, C.c_uint64]
    g("session_free").argtypes = [C.c_void_p]
    g("session_index_size").restype = C.c_int64; g("session_index_size").argtypes = [C.c_void_p]
    g("session_index_copy").argtypes = [C.c_void_p, C.POINTER(Minmer)]
    g("session_nkeys").restype = C.c_int64; g("session_nkeys").argtypes = [C.c_void_p]
    g("session_keys").argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
    g("session_lookup").restype = C.c_int64
    g("session_lookup").argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Point), C.c_int64]
    g("session_is_freq").restype = C.c_int; g("session_is_freq").argtypes = [C.c_void_p, C.c_uint64]
    g("session_freq_threshold").restype = C.c_int; g("session_freq_threshold").argtypes = [C.c_void_p]
    g("session_ncontigs").restype = C.c_int; g("session_ncontigs").argtypes = [C.c_void_p]
    g("session_contig_len").restype = C.c_int; g("session_contig_len").argtypes = [C.c_void_p, C.c_int]
    g("session_ncutoffs").restype = C.c_int; g("session_ncutoffs").argtypes = [C.c_void_p]
    g("session_cutoffs").argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    g("session_map_fragment").restype = C.c_int
    g("session_map_fragment").argtypes = [
        C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p,
        C.POINTER(Minmer), C.c_int, C.POINTER(Point), C.c_int, C.POINTER(L1), C.c_int,
        C.POINTER(L2), C.POINTER(C.c_int), C.c_int, C.POINTER(Mapping), C.c_int,
        C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    g("session_map_read").restype = C.c_int
    g("session_map_read").argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(Mapping), C.c_int]


class _Side:
    """uniform python face over liboracle.so (prefix orc_) and libmashmap_ref.so (prefix ref_)"""

    def __init__(self, lib, prefix):
        self.lib, self.p = lib, prefix
        _bind_common(lib, prefix)

    def f(self, name):
        return getattr(self.lib, self.p + name)

    def get_hash(self, s):
        return int(self.f("get_hash")(s, len(s)))

    def sketch_sequence(self, seq, k, s, seqId=0):
        b = bytes(seq)
        buf = (Minmer * (s + 1))()
        n = self.f("sketch_sequence")(b, len(b), k, s, seqId, buf, s + 1)
        return [buf[i].key() for i in range(n)]

    def add_minmers(self, seq, k, w, s, seqId=0):
        b = bytes(seq)
        cap = max(1024, len(b))
        buf = (Minmer * cap)()
        n = self.f("add_minmers")(b, len(b), k, w, s, seqId, buf, cap)
        assert n <= cap
        return np.frombuffer(buf, dtype=MINMER_DT, count=n).copy()

    # -- sessions
    def index_array(self, h):
        n = self.f("session_index_size")(h)
        buf = (Minmer * max(1, n))()
        self.f("session_index_copy")(h, buf)
        return np.frombuffer(buf, dtype=MINMER_DT, count=n).copy()

    def keys(self, h):
        n = self.f("session_nkeys")(h)
        k = np.zeros(n, dtype=np.uint64); c = np.zeros(n, dtype=np.int64)
        self.f("session_keys")(h, k.ctypes.data_as(C.POINTER(C.c_uint64)), c.ctypes.data_as(C.POINTER(C.c_int64)))
        o = np.argsort(k, kind="stable")
        return k[o], c[o]

    def lookup(self, h, hash_):
        cap = 4096
        while True:
            buf = (Point * cap)()
            n = self.f("session_lookup")(h, int(hash_), buf, cap)
            if n <= cap:
                break
            cap = n
        if n < 0:
            return None
        return [buf[i].key() for i in range(n)]

    def cutoffs(self, h):
        n = self.f("session_ncutoffs")(h)
        buf = (C.c_int * n)()
        self.f("session_cutoffs")(h, buf)
        return list(buf)

    def map_fragment(self, h, seq, seqCounter=0, name=b"q", fullLen=None, s=1024):
        b = bytes(seq)
        qsk = (Minmer * (s + 1))(); pts = (Point * 65536)(); l1 = (L1 * 1024)(); l2 = (L2 * 4096)()
        l2c = (C.c_int * 4096)(); maps = (Mapping * 1024)(); counts = (C.c_int64 * 8)(); kc = C.c_double(0)
        self.f("session_map_fragment")(h, b, len(b), fullLen if fullLen is not None else len(b), seqCounter, name,
                                       qsk, s + 1, pts, 65536, l1, 1024, l2, l2c, 4096, maps, 1024, counts, C.byref(kc))
        c = list(counts)
        assert c[1] <= 65536 and c[2] <= 1024 and c[3] <= 4096 and c[4] <= 1024
        return dict(
            sketch=[qsk[i].key() for i in range(c[0])],
            points=[pts[i].key() for i in range(c[1])],
            l1=[l1[i].key() for i in range(c[2])],
            l2=[(l2c[i],) + l2[i].key() for i in range(c[3])],
            maps_i=[maps[i].ikey() for i in range(c[4])],
            maps_f=[maps[i].fkey() for i in range(c[4])],
            minimumHits=c[5], sketchSize=c[6], rawSketchSize=c[7], kmerComplexity=kc.value)


MINMER_DT = np.dtype([("hash", "<u8"), ("wpos", "<i4"), ("wpos_end", "<i4"), ("seqId", "<i4"), ("strand", "<i2"),
                      ("pad", "<i2")])
POINT_DT = np.dtype([("pos", "<i4"), ("pad0", "<i4"), ("hash", "<u8"), ("seqId", "<i4"), ("side", "i1"),
                     ("pad1", "i1", (3,))])


class Oracle(_Side):
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = C.CDLL(ORACLE_SO)
        super().__init__(lib, "orc_")
        lib.orc_session_new.restype = C.c_void_p
        lib.orc_session_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_char, C.c_float, C.c_int]
        lib.orc_session_add_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        lib.orc_session_finalize.argtypes = [C.c_void_p]
        lib.orc_normalise.argtypes = [C.c_char_p, C.c_int64]

    def session(self, contigs, k=19, segLength=5000, sketchSize=130, pi=0.85, filterMode=FILTER_MAP, flags=FLAG_HG,
                delim=b"\0", kmerPct=0.001, numMappings=1, mutate_index=None):
        """mutate_index(records) -> records: replaces minmerIndex before Sketch::index runs (an index as another program might
        have written it)"""
        h = self.lib.orc_session_new(k, segLength, sketchSize, pi, filterMode, flags, delim, kmerPct, numMappings)
        for name, a in contigs:
            b = bytes(a)
            self.lib.orc_session_add_contig(h, name.encode(), b, len(b))
        if mutate_index is not None:
            recs = np.ascontiguousarray(mutate_index(self.index_array(h)), dtype=MINMER_DT)
            self.lib.orc_session_set_index.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
            self.lib.orc_session_set_index(h, recs.ctypes.data, len(recs))
        self.lib.orc_session_finalize(h)
        return h

    def free(self, h):
        self.lib.orc_session_free(h)

    def export_index(self, h):
        """everything mm_index_upload needs, as numpy arrays in the C-ABI layouts"""
        L = self.lib
        L.orc_session_npoints.restype = C.c_int64; L.orc_session_npoints.argtypes = [C.c_void_p]
        L.orc_session_nfreq.restype = C.c_int64; L.orc_session_nfreq.argtypes = [C.c_void_p]
        L.orc_session_export_lookup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_session_freq_list.argtypes = [C.c_void_p, C.c_void_p]
        nk = L.orc_session_nkeys(h); npt = L.orc_session_npoints(h); nf = L.orc_session_nfreq(h)
        keys = np.zeros(nk, dtype=np.uint64); offs = np.zeros(nk + 1, dtype=np.uint64); pts = np.zeros(npt, dtype=POINT_DT)
        L.orc_session_export_lookup(h, keys.ctypes.data, offs.ctypes.data, pts.ctypes.data)
        freq = np.zeros(nf, dtype=np.uint64)
        if nf:
            L.orc_session_freq_list(h, freq.ctypes.data)
        nc = L.orc_session_ncontigs(h)
        clen = np.array([L.orc_session_contig_len(h, i) for i in range(nc)], dtype=np.int32)
        return dict(minmers=self.index_array(h), keys=keys, offsets=offs, points=pts, freq=freq, contigLen=clen)

    def min_hits_table(self, s, k, pi):
        return np.array([0] + [self.lib.orc_min_hits_relaxed(q, k, pi) for q in range(1, s + 1)], dtype=np.int32)


class Ref(_Side):
    """the real reference; only available where oracle/_ref has been built"""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        lib = C.CDLL(REF_SO)
        super().__init__(lib, "ref_")
        lib.ref_session_new.restype = C.c_void_p
        lib.ref_session_new.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_char,
                                        C.c_float, C.c_int, C.c_int]

    def session(self, fasta_paths, k=19, segLength=5000, sketchSize=130, pi=0.85, filterMode=FILTER_MAP, flags=FLAG_HG,
                delim=b"\0", kmerPct=0.001, numMappings=1, threads=1):
        return self.lib.ref_session_new("\n".join(fasta_paths).encode(), k, segLength, sketchSize, pi, filterMode,
                                        flags, delim, kmerPct, numMappings, threads)

    def free(self, h):
        self.lib.ref_session_free(h)
