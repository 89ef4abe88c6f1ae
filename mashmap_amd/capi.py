"""ctypes binding of libmashmap_hip.so (include/mashmap_hip.h).

Plumbing only: the product is the HIP library; the C++ host side lives in mashmap_amd/host/.
There is no CPU fallback here -- if the library is missing or no gfx950 device is present every
call fails loudly (LibraryMissing / MashmapError).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libmashmap_hip.so")

MM_FLAG_HG_FILTER, MM_FLAG_SKIP_SELF, MM_FLAG_SKIP_PREFIX, MM_FLAG_LOWER_TRIANGULAR, MM_FLAG_NO_SPLIT = 1, 2, 4, 8, 16
KERNELS = ["pack", "sketch", "sketch_hard", "lookup", "sort", "l1", "l2", "refhash", "l2_locate", "winnow", "select"]


class LibraryMissing(RuntimeError):
    pass


class MashmapError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("kmerSize", C.c_int32), ("segLength", C.c_int32), ("sketchSize", C.c_int32), ("flags", C.c_int32)]


MINMER_DT = np.dtype([("hash", "<u8"), ("wpos", "<i4"), ("wpos_end", "<i4"), ("seqId", "<i4"), ("strand", "<i2"),
                      ("pad", "<i2")])
POINT_DT = np.dtype([("pos", "<i4"), ("pad0", "<i4"), ("hash", "<u8"), ("seqId", "<i4"), ("side", "i1"),
                     ("pad1", "i1", (3,))])
FRAG_DT = np.dtype([("readId", "<i4"), ("fragStart", "<i4"), ("len", "<i4"), ("pad", "<i4")])
STATS_DT = np.dtype([("rawSketchSize", "<i4"), ("sketchSize", "<i4"), ("maxHash", "<u8"), ("nPoints", "<i4"),
                     ("nL1", "<i4")])
L1_DT = np.dtype([("frag", "<i4"), ("seqId", "<i4"), ("rangeStartPos", "<i4"), ("rangeEndPos", "<i4"),
                  ("intersectionSize", "<i4")])
MAPPING_DT = np.dtype([("querySeqId", "<i4"), ("fragStart", "<i4"), ("fragLen", "<i4"), ("refSeqId", "<i4"), ("refStartPos", "<i4"),
                       ("conservedSketches", "<i4"), ("sketchSize", "<i4"), ("strand", "<i4"), ("rawSketchSize", "<i4"), ("pad", "<i4"),
                       ("maxHash", "<u8")])
COMM_ID_BYTES = 128
L2_DT = np.dtype([("frag", "<i4"), ("cand", "<i4"), ("seqId", "<i4"), ("meanOptimalPos", "<i4"),
                  ("optimalStart", "<i4"), ("optimalEnd", "<i4"), ("sharedSketchSize", "<i4"), ("strand", "<i4")])
assert MAPPING_DT.itemsize == 48 and MINMER_DT.itemsize == 24 and POINT_DT.itemsize == 24 and STATS_DT.itemsize == 24 and L2_DT.itemsize == 32

_lib = None


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def load():
    """dlopen the in-tree library; raises LibraryMissing when it has not been built"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(or make -C mashmap_amd/csrc)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, sz, i32, i64 = C.c_void_p, C.c_size_t, C.c_int32, C.c_int64
    sig = {
        "mm_abi_version": (C.c_int, []),
        "mm_create": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(Params)]),
        "mm_destroy": (None, [vp]),
        "mm_last_error": (C.c_char_p, [vp]),
        "mm_index_upload": (C.c_int, [vp, vp, sz, vp, vp, sz, vp, sz, vp, sz, vp, vp, sz]),
        "mm_set_tables": (C.c_int, [vp, vp, sz, vp, sz]),
        "mm_set_tables_default": (C.c_int, [vp, C.c_float]),
        "mm_stat_j2md": (C.c_float, [C.c_float, C.c_int]),
        "mm_stat_md2j": (C.c_float, [C.c_float, C.c_int]),
        "mm_stat_md_lower_bound": (C.c_float, [C.c_float, C.c_int, C.c_int, C.c_float]),
        "mm_stat_min_hits_relaxed": (C.c_int, [C.c_int, C.c_int, C.c_float]),
        "mm_stat_recommended_sketch_size": (i64, [C.c_int, C.c_float, i64, C.c_uint64]),
        "mm_stat_sketch_cutoffs": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, sz]),
        "mm_stat_replay_tables": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, vp, vp]),
        "mm_results_copy_device": (C.c_int, [vp, vp, sz, C.POINTER(sz)]),
        "mm_reads_upload": (C.c_int, [vp, vp, vp, sz, vp, vp, i32]),
        "mm_reads_upload_device": (C.c_int, [vp, vp, sz, vp, sz, vp, vp, i32]),
        "mm_num_fragments": (sz, [vp]),
        "mm_host_alloc": (vp, [sz]),
        "mm_host_free": (None, [vp]),
        "mm_fragments_download": (C.c_int, [vp, vp]),
        "mm_sketch_fragments": (C.c_int, [vp]),
        "mm_sketch_download": (C.c_int, [vp, vp, vp]),
        "mm_map_fragments": (C.c_int, [vp]),
        "mm_result_counts": (C.c_int, [vp, C.POINTER(sz), C.POINTER(sz)]),
        "mm_results_download": (C.c_int, [vp, vp, vp, vp]),
        "mm_query_sketch_download": (C.c_int, [vp, vp]),
        "mm_points_download": (C.c_int, [vp, sz, vp, sz, C.POINTER(sz)]),
        "mm_results_device": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "mm_index_build": (C.c_int, [vp, vp, vp, sz, vp, C.c_float]),
        "mm_index_sizes": (C.c_int, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(i32)]),
        "mm_index_download": (C.c_int, [vp, vp, vp, vp, vp, vp]),
        "mm_set_option": (C.c_int, [vp, C.c_int, C.c_int]),
        "mm_index_download_full": (C.c_int, [vp, vp, C.POINTER(sz)]),
        "mm_index_upload_full": (C.c_int, [vp, vp, sz, vp, vp, sz, vp, sz, vp, vp, sz, C.c_float]),
        "mm_profile_enable": (C.c_int, [vp, C.c_int]),
        "mm_profile_read": (C.c_int, [vp, vp, vp, C.c_int]),
        "mm_kernel_name": (C.c_char_p, [C.c_int]),
        "mm_bench_hash_only": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double)]),
        "mm_set_replay_tables": (C.c_int, [vp, vp, vp, sz]),
        "mm_mappings_count": (C.c_int, [vp, C.POINTER(sz)]),
        "mm_mappings_download": (C.c_int, [vp, vp, sz, C.POINTER(sz)]),
        "mm_mappings_device": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "mm_comm_unique_id": (C.c_int, [vp]),
        "mm_comm_init_rank": (C.c_int, [vp, vp, C.c_int, C.c_int]),
        "mm_comm_init_local": (C.c_int, [C.POINTER(vp), C.c_int]),
        "mm_comm_world": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "mm_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.c_char_p, sz]),
        "mm_allgatherv_mappings": (C.c_int, [vp]),
        "mm_allgatherv_mappings_local": (C.c_int, [C.POINTER(vp), C.c_int]),
        "mm_allgatherv_mappings_begin": (C.c_int, [vp]),
        "mm_allgatherv_mappings_end": (C.c_int, [vp]),
        "mm_gathered_counts": (C.c_int, [vp, vp, C.POINTER(sz)]),
        "mm_gathered_download": (C.c_int, [vp, vp, sz]),
        "mm_gathered_device": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "mm_index_replicate": (C.c_int, [vp, vp]),
        "mm_reads_prefetch": (C.c_int, [vp, vp, sz]),
        "mm_reads_upload_packed": (C.c_int, [vp, vp, vp, vp, vp, vp, sz, vp, vp, i32]),
        "mm_reads_prefetch_packed": (C.c_int, [vp, vp, vp, sz]),
        "mm_pack_read": (sz, [vp, sz, vp, vp]),
        "mm_pack_read_portable": (sz, [vp, sz, vp, vp]),
        "mm_reads_packed_download": (C.c_int, [vp, vp, vp, vp, C.POINTER(sz)]),
        "mm_index_layout_get": (C.c_int, [vp, vp]),
        "mm_pass_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int), vp]),
        "mm_pass_totals": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "mm_reads_exchange": (C.c_int, [vp, C.c_int]),
        "mm_reads_upload_packed_parts": (C.c_int, [vp, vp, sz, C.c_int32]),
        "mm_reads_prefetch_packed_append": (C.c_int, [vp, vp, vp, sz, sz, C.POINTER(C.c_int)]),
        "mm_reads_prefetch_drop": (C.c_int, [vp]),
        "mm_reads_prefetch_reserve": (C.c_int, [vp, sz]),
        "mm_synchronize": (C.c_int, [vp]),
        "mm_stream": (vp, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here == header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTS = ["mm_abi_version", "mm_create", "mm_destroy", "mm_last_error", "mm_index_upload", "mm_set_tables",
           "mm_set_tables_default", "mm_stat_j2md", "mm_stat_md2j", "mm_stat_md_lower_bound", "mm_stat_min_hits_relaxed",
           "mm_stat_recommended_sketch_size", "mm_stat_sketch_cutoffs", "mm_results_copy_device",
           "mm_reads_upload", "mm_reads_upload_device", "mm_num_fragments", "mm_fragments_download",
           "mm_sketch_fragments", "mm_sketch_download", "mm_map_fragments", "mm_result_counts",
           "mm_results_download", "mm_query_sketch_download", "mm_points_download", "mm_results_device",
           "mm_index_build", "mm_index_sizes", "mm_index_download", "mm_set_option", "mm_index_download_full", "mm_index_upload_full", "mm_profile_enable", "mm_profile_read",
           "mm_kernel_name", "mm_synchronize", "mm_stream", "mm_bench_hash_only",
           "mm_set_replay_tables", "mm_mappings_count", "mm_mappings_download", "mm_mappings_device", "mm_comm_unique_id",
           "mm_comm_init_rank", "mm_comm_init_local", "mm_comm_world", "mm_allgatherv_mappings", "mm_allgatherv_mappings_local",
           "mm_allgatherv_mappings_begin", "mm_allgatherv_mappings_end",
           "mm_gathered_counts", "mm_gathered_download", "mm_gathered_device", "mm_index_replicate", "mm_stat_replay_tables", "mm_host_alloc", "mm_host_free", "mm_reads_prefetch",
           "mm_reads_upload_packed", "mm_reads_prefetch_packed", "mm_pack_read", "mm_pack_read_portable", "mm_reads_packed_download",
           "mm_index_layout_get", "mm_pass_stats", "mm_comm_info", "mm_pass_totals", "mm_reads_exchange", "mm_reads_upload_packed_parts",
           "mm_reads_prefetch_packed_append", "mm_reads_prefetch_drop", "mm_reads_prefetch_reserve"]


def stat_sketch_cutoffs(sketchSize, k, hg=True):
    lib = load()
    n = lib.mm_stat_sketch_cutoffs(sketchSize, k, 1 if hg else 0, None, 0)
    out = np.zeros(n, dtype=np.int32)
    lib.mm_stat_sketch_cutoffs(sketchSize, k, 1 if hg else 0, _ptr(out), n)
    return out


def comm_unique_id():
    buf = (C.c_char * COMM_ID_BYTES)()
    if load().mm_comm_unique_id(C.cast(buf, C.c_void_p)) != 0:
        raise MashmapError("mm_comm_unique_id failed (RCCL not loadable?)")
    return bytes(buf)


def comm_init_local(ctxs):
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    if load().mm_comm_init_local(arr, len(ctxs)) != 0:
        raise MashmapError("mm_comm_init_local failed: " + load().mm_last_error(ctxs[0].h).decode())


def allgatherv_mappings_local(ctxs):
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    if load().mm_allgatherv_mappings_local(arr, len(ctxs)) != 0:
        raise MashmapError("mm_allgatherv_mappings_local failed: " + load().mm_last_error(ctxs[0].h).decode())


def stat_replay_tables(sketchSize, k, pi, aniDiff=0.0, keepLow=True):
    n = (sketchSize + 1) ** 2
    acc = np.zeros(n, dtype=np.uint8); mi = np.zeros(n, dtype=np.int16)
    if load().mm_stat_replay_tables(sketchSize, k, pi, aniDiff, 1 if keepLow else 0, _ptr(acc), _ptr(mi)) != 0:
        raise MashmapError("mm_stat_replay_tables failed")
    return acc, mi


def pack_reads(reads, portable=False):
    """host-side makeUpperCaseAndValidDNA + 2-bit packing of a list of uint8 arrays (mm_pack_read; no GPU needed): the arrays
    mm_reads_upload_packed takes -- (bases2 uint32, nmask uint32, hasN uint8, lengths int32)"""
    lib = load()
    fn = lib.mm_pack_read_portable if portable else lib.mm_pack_read
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    groups = (lens.astype(np.int64) + 31) // 32
    start = np.concatenate([[0], np.cumsum(groups)])
    b2 = np.zeros(max(1, int(start[-1]) * 2), dtype=np.uint32); nm = np.zeros(max(1, int(start[-1])), dtype=np.uint32)
    hasn = np.zeros(max(1, len(reads)), dtype=np.uint8)
    for i, r in enumerate(reads):
        a = np.ascontiguousarray(r, dtype=np.uint8)
        nN = fn(_ptr(a), len(a), C.c_void_p(b2.ctypes.data + 8 * int(start[i])), C.c_void_p(nm.ctypes.data + 4 * int(start[i])))
        hasn[i] = 1 if nN else 0
    return b2[:int(start[-1]) * 2] if start[-1] else b2[:0], nm[:int(start[-1])] if start[-1] else nm[:0], hasn[:len(reads)], lens


class PackedPart(C.Structure):
    """mm_packed_part (include/mashmap_hip.h)"""
    _fields_ = [("bases2", C.c_void_p), ("nmask", C.c_void_p), ("readHasN", C.c_void_p), ("readLengths", C.c_void_p), ("readStarts", C.c_void_p),
                ("nReads", C.c_size_t), ("readRefGroup", C.c_void_p), ("readSelfSeqId", C.c_void_p)]


class Context:
    """one mm_ctx (one GPU)"""

    def __init__(self, k=19, segLength=5000, sketchSize=130, flags=MM_FLAG_HG_FILTER, device=0):
        self.lib = load()
        self.params = Params(k, segLength, sketchSize, flags)
        h = C.c_void_p()
        rc = self.lib.mm_create(C.byref(h), device, C.byref(self.params))
        if rc != 0:
            raise MashmapError("mm_create failed (%d): %s" % (rc, self.lib.mm_last_error(None).decode()))
        self.h = h
        self.s = sketchSize

    def close(self):
        if getattr(self, "h", None):
            self.lib.mm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise MashmapError("%s failed (%d): %s" % (what, rc, self.lib.mm_last_error(self.h).decode()))

    # ---- index
    def index_upload(self, minmers, keys, offsets, points, freq, contigLen, refGroup=None):
        minmers = np.ascontiguousarray(minmers, dtype=MINMER_DT); keys = np.ascontiguousarray(keys, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64); points = np.ascontiguousarray(points, dtype=POINT_DT)
        freq = np.ascontiguousarray(freq, dtype=np.uint64); contigLen = np.ascontiguousarray(contigLen, dtype=np.int32)
        rg = np.ascontiguousarray(refGroup, dtype=np.int32) if refGroup is not None else None
        self._keep = (minmers, keys, offsets, points, freq, contigLen, rg)
        self._ck(self.lib.mm_index_upload(self.h, _ptr(minmers), len(minmers), _ptr(keys), _ptr(offsets), len(keys),
                                          _ptr(points), len(points), _ptr(freq), len(freq), _ptr(contigLen), _ptr(rg),
                                          len(contigLen)), "mm_index_upload")

    def index_build(self, contigs, refGroup=None, kmerPct=0.001):
        """contigs: list of uint8 arrays (ASCII)"""
        offs = np.zeros(len(contigs) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(c) for c in contigs])
        # contigs that are consecutive views of one array (bench.py lays its reference out that way) are passed as they lie
        ptr = [c.ctypes.data for c in contigs]
        if len(contigs) and all(c.dtype == np.uint8 and c.flags["C_CONTIGUOUS"] for c in contigs) and all(ptr[i] + len(contigs[i]) == ptr[i + 1] for i in range(len(contigs) - 1)):
            buf = np.ctypeslib.as_array(C.cast(ptr[0], C.POINTER(C.c_uint8)), shape=(int(offs[-1]),)) if offs[-1] else np.zeros(0, dtype=np.uint8)
            self._keep = contigs                       # (the views own the memory for the duration of the call)
        else:
            buf = np.concatenate(contigs) if len(contigs) else np.zeros(0, dtype=np.uint8)
        rg = np.ascontiguousarray(refGroup, dtype=np.int32) if refGroup is not None else None
        self._ck(self.lib.mm_index_build(self.h, _ptr(buf), _ptr(offs), len(contigs), _ptr(rg), kmerPct), "mm_index_build")

    def index_layout(self):
        """mm_index_layout_get as a dict (seedTableSlots, seedTableBytes, tagBytes, filterBytes, events, openRecords, tagged)"""
        a = np.zeros(7, dtype=np.uint64)
        self._ck(self.lib.mm_index_layout_get(self.h, _ptr(a)), "mm_index_layout_get")
        d = dict(zip(("seedTableSlots", "seedTableBytes", "tagBytes", "filterBytes", "events", "openRecords"), (int(x) for x in a[:6])))
        d["tagged"] = int(a[6] & 0xFFFFFFFF)
        return d

    def index_download(self):
        n = [C.c_size_t() for _ in range(4)]; ft = C.c_int32()
        self._ck(self.lib.mm_index_sizes(self.h, *[C.byref(x) for x in n], C.byref(ft)), "mm_index_sizes")
        nm, nk, npt, nf = [x.value for x in n]
        mins = np.zeros(nm, dtype=MINMER_DT); keys = np.zeros(nk, dtype=np.uint64); offs = np.zeros(nk + 1, dtype=np.uint64)
        pts = np.zeros(npt, dtype=POINT_DT); fr = np.zeros(nf, dtype=np.uint64)
        self._ck(self.lib.mm_index_download(self.h, _ptr(mins), _ptr(keys), _ptr(offs), _ptr(pts), _ptr(fr)), "mm_index_download")
        return dict(minmers=mins, keys=keys, offsets=offs, points=pts, freq=fr, freqThreshold=ft.value)

    def set_tables(self, minHits, cutoffs):
        a = np.ascontiguousarray(minHits, dtype=np.int32); b = np.ascontiguousarray(cutoffs, dtype=np.int32)
        self._ck(self.lib.mm_set_tables(self.h, _ptr(a), len(a), _ptr(b), len(b)), "mm_set_tables")

    def set_tables_default(self, pi):
        self._ck(self.lib.mm_set_tables_default(self.h, pi), "mm_set_tables_default")

    def results_copy_device(self, dptr, cap):
        n = C.c_size_t()
        self._ck(self.lib.mm_results_copy_device(self.h, C.c_void_p(dptr), cap, C.byref(n)), "mm_results_copy_device")
        return n.value

    # ---- reads
    def reads_upload(self, reads, refGroup=None, selfSeqId=None, seqCounterBase=0):
        """reads: list of uint8 arrays (ASCII) or (concatenated uint8 array, int64 offsets)"""
        if isinstance(reads, tuple):
            buf, offs = reads
            buf = np.ascontiguousarray(buf, dtype=np.uint8); offs = np.ascontiguousarray(offs, dtype=np.int64)
        else:
            offs = np.zeros(len(reads) + 1, dtype=np.int64)
            offs[1:] = np.cumsum([len(r) for r in reads])
            buf = np.concatenate(reads) if len(reads) else np.zeros(0, dtype=np.uint8)
        n = len(offs) - 1
        rg = np.ascontiguousarray(refGroup, dtype=np.int32) if refGroup is not None else None
        ss = np.ascontiguousarray(selfSeqId, dtype=np.int32) if selfSeqId is not None else None
        self._ck(self.lib.mm_reads_upload(self.h, _ptr(buf), _ptr(offs), n, _ptr(rg), _ptr(ss), seqCounterBase), "mm_reads_upload")
        self._nreads = n
        return self.num_fragments()

    def reads_upload_packed(self, packed, refGroup=None, selfSeqId=None, seqCounterBase=0, prefetch=False, starts=None):
        """packed: what pack_reads() returns (bases2, nmask, hasN, lengths); prefetch=True sends the words ahead with mm_reads_prefetch_packed;
        starts: packed base of every read's first base (gapped layout), None = the reads follow each other"""
        b2, nm, hasn, lens = packed
        st = np.ascontiguousarray(starts, dtype=np.int64) if starts is not None else None
        rg = np.ascontiguousarray(refGroup, dtype=np.int32) if refGroup is not None else None
        ss = np.ascontiguousarray(selfSeqId, dtype=np.int32) if selfSeqId is not None else None
        if prefetch:
            self._ck(self.lib.mm_reads_prefetch_packed(self.h, _ptr(b2), _ptr(nm), nm.size * 32), "mm_reads_prefetch_packed")
        self._ck(self.lib.mm_reads_upload_packed(self.h, _ptr(b2), _ptr(nm), _ptr(hasn), _ptr(lens), _ptr(st), len(lens), _ptr(rg), _ptr(ss), seqCounterBase),
                 "mm_reads_upload_packed")
        self._nreads = len(lens)
        return self.num_fragments()

    def reads_upload_packed_parts(self, parts, seqCounterBase=0, stage=()):
        """parts: list of dicts(packed=(bases2, nmask, hasN, lengths), starts=None, refGroup=None, selfSeqId=None) laid end to end as one
        resident batch (mm_reads_upload_packed_parts); stage: indices of the parts sent ahead first with mm_reads_prefetch_packed_append"""
        arr = (PackedPart * len(parts))()
        keep = []
        for i, p in enumerate(parts):
            b2, nm, hasn, lens = p["packed"]
            st = np.ascontiguousarray(p["starts"], dtype=np.int64) if p.get("starts") is not None else None
            rg = np.ascontiguousarray(p["refGroup"], dtype=np.int32) if p.get("refGroup") is not None else None
            ss = np.ascontiguousarray(p["selfSeqId"], dtype=np.int32) if p.get("selfSeqId") is not None else None
            keep.append((b2, nm, hasn, lens, st, rg, ss))
            arr[i] = PackedPart(_ptr(b2), _ptr(nm), _ptr(hasn), _ptr(lens), _ptr(st), len(lens), _ptr(rg), _ptr(ss))
        reserve = sum(k[1].size * 32 for k in keep)
        for i in stage:
            self._ck(self.lib.mm_reads_prefetch_packed_append(self.h, _ptr(keep[i][0]), _ptr(keep[i][1]), keep[i][1].size * 32, reserve, None), "mm_reads_prefetch_packed_append")
        self._ck(self.lib.mm_reads_upload_packed_parts(self.h, arr, len(parts), seqCounterBase), "mm_reads_upload_packed_parts")
        self._nreads = sum(len(k[3]) for k in keep)
        return self.num_fragments()

    def reads_packed_download(self):
        n = C.c_size_t()
        self._ck(self.lib.mm_reads_packed_download(self.h, None, None, None, C.byref(n)), "mm_reads_packed_download")
        b2 = np.zeros(n.value // 16, dtype=np.uint32); nm = np.zeros(n.value // 32, dtype=np.uint32)
        nreads = C.c_size_t()
        hasn = np.zeros(max(1, self._nreads), dtype=np.uint32)
        self._ck(self.lib.mm_reads_packed_download(self.h, _ptr(b2), _ptr(nm), _ptr(hasn), C.byref(n)), "mm_reads_packed_download")
        return b2, nm, hasn[:self._nreads]

    def reads_prefetch(self, buf):
        """start the H2D copy of the concatenated uint8 array the next reads_upload((buf, offs)) will pass (the array must stay alive
        and unchanged until then)"""
        assert buf.dtype == np.uint8 and buf.flags["C_CONTIGUOUS"]
        self._ck(self.lib.mm_reads_prefetch(self.h, _ptr(buf), buf.size), "mm_reads_prefetch")

    def reads_upload_device(self, dptr, nbytes, offs, refGroup=None, selfSeqId=None, seqCounterBase=0):
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        rg = np.ascontiguousarray(refGroup, dtype=np.int32) if refGroup is not None else None
        ss = np.ascontiguousarray(selfSeqId, dtype=np.int32) if selfSeqId is not None else None
        self._ck(self.lib.mm_reads_upload_device(self.h, C.c_void_p(dptr), nbytes, _ptr(offs), len(offs) - 1, _ptr(rg), _ptr(ss),
                                                 seqCounterBase), "mm_reads_upload_device")
        self._nreads = len(offs) - 1
        return self.num_fragments()

    def num_fragments(self):
        return int(self.lib.mm_num_fragments(self.h))

    def fragments(self):
        out = np.zeros(self.num_fragments(), dtype=FRAG_DT)
        self._ck(self.lib.mm_fragments_download(self.h, _ptr(out)), "mm_fragments_download")
        return out

    # ---- kernels
    def sketch(self):
        self._ck(self.lib.mm_sketch_fragments(self.h), "mm_sketch_fragments")
        nF = self.num_fragments()
        out = np.zeros((nF, self.s), dtype=MINMER_DT); cnt = np.zeros(nF, dtype=np.uint32)
        self._ck(self.lib.mm_sketch_download(self.h, _ptr(out), _ptr(cnt)), "mm_sketch_download")
        return out, cnt

    def sketch_only(self):
        self._ck(self.lib.mm_sketch_fragments(self.h), "mm_sketch_fragments")

    def map(self):
        self._ck(self.lib.mm_map_fragments(self.h), "mm_map_fragments")

    def pass_stats(self):
        """(host synchronisations inside the last map(), whether it was a steady-state pass)"""
        n = C.c_uint64(); st = C.c_int()
        self._ck(self.lib.mm_pass_stats(self.h, C.byref(n), C.byref(st), None), "mm_pass_stats")
        return int(n.value), bool(st.value)

    def pass_totals(self):
        """dict(passes, steady, redone): this context's map() calls so far, how many went through as steady-state passes, and how many
        steady-state attempts outgrew a buffer and were redone the sized way"""
        a, b, c_ = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._ck(self.lib.mm_pass_totals(self.h, C.byref(a), C.byref(b), C.byref(c_)), "mm_pass_totals")
        return {"passes": int(a.value), "steady": int(b.value), "redone": int(c_.value)}

    def reads_exchange(self, slot):
        """swaps the resident batch of reads with the one parked in `slot` (0 .. MM_BATCH_SLOTS - 1); no copy"""
        self._ck(self.lib.mm_reads_exchange(self.h, slot), "mm_reads_exchange")

    def pass_counts(self):
        """dict(l1, l2, queued, stream_entries, hard) of the last map()"""
        a = np.zeros(5, dtype=np.uint64)
        self._ck(self.lib.mm_pass_stats(self.h, None, None, _ptr(a)), "mm_pass_stats")
        return dict(zip(("l1", "l2", "queued", "stream_entries", "hard"), (int(x) for x in a)))

    def results(self):
        n1, n2 = C.c_size_t(), C.c_size_t()
        self._ck(self.lib.mm_result_counts(self.h, C.byref(n1), C.byref(n2)), "mm_result_counts")
        nF = self.num_fragments()
        stats = np.zeros(nF, dtype=STATS_DT); l1 = np.zeros(n1.value, dtype=L1_DT); l2 = np.zeros(n2.value, dtype=L2_DT)
        self._ck(self.lib.mm_results_download(self.h, _ptr(stats), _ptr(l1), _ptr(l2)), "mm_results_download")
        return stats, l1, l2

    def result_counts(self):
        n1, n2 = C.c_size_t(), C.c_size_t()
        self._ck(self.lib.mm_result_counts(self.h, C.byref(n1), C.byref(n2)), "mm_result_counts")
        return n1.value, n2.value

    def results_device(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.mm_results_device(self.h, C.byref(p), C.byref(n)), "mm_results_device")
        return p.value, n.value

    def query_sketches(self):
        out = np.zeros((self.num_fragments(), self.s), dtype=MINMER_DT)
        self._ck(self.lib.mm_query_sketch_download(self.h, _ptr(out)), "mm_query_sketch_download")
        return out

    def reserve_fragments(self, n):
        """MM_OPT_RESERVE_FRAGMENTS: the next sized pass sizes every staging buffer for a batch of n fragments"""
        self._ck(self.lib.mm_set_option(self.h, 3, int(n)), "mm_set_option")

    def keep_points(self, on=True):
        """MM_OPT_KEEP_POINTS: keep every fragment's sorted interval points in HBM (needed by points()); on=2: ... after the interval-point
        pre-filter of the HBM point path (k_filter_points) has run on them"""
        self._ck(self.lib.mm_set_option(self.h, 1, int(on) if on in (0, 1, 2) else 1), "mm_set_option")

    def points(self, frag, cap=1 << 16):
        out = np.zeros(cap, dtype=POINT_DT); n = C.c_size_t()
        self._ck(self.lib.mm_points_download(self.h, frag, _ptr(out), cap, C.byref(n)), "mm_points_download")
        return out[:n.value]

    # ---- profiling
    def profile(self, on=True):
        self.lib.mm_profile_enable(self.h, 1 if on else 0)

    def profile_read(self, reset=True):
        ms = np.zeros(len(KERNELS), dtype=np.float64); ln = np.zeros(len(KERNELS), dtype=np.uint64)
        self.lib.mm_profile_read(self.h, _ptr(ms), _ptr(ln), 1 if reset else 0)
        return {KERNELS[i]: (float(ms[i]), int(ln[i])) for i in range(len(KERNELS))}

    # ---- candidate mappings + multi-GPU exchange
    def set_replay_tables(self, accept, minIsz):
        a = np.ascontiguousarray(accept, dtype=np.uint8); b = np.ascontiguousarray(minIsz, dtype=np.int16)
        self._ck(self.lib.mm_set_replay_tables(self.h, _ptr(a), _ptr(b), self.s + 1), "mm_set_replay_tables")

    def mappings(self):
        n = C.c_size_t()
        self._ck(self.lib.mm_mappings_count(self.h, C.byref(n)), "mm_mappings_count")
        out = np.zeros(n.value, dtype=MAPPING_DT)
        self._ck(self.lib.mm_mappings_download(self.h, _ptr(out), n.value, C.byref(n)), "mm_mappings_download")
        return out

    def comm_init_rank(self, comm_id, rank, world):
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        self._ck(self.lib.mm_comm_init_rank(self.h, C.cast(buf, C.c_void_p), rank, world), "mm_comm_init_rank")

    def comm_info(self):
        """dict(world_seen, library_path) of this context's communicator"""
        n = C.c_int(); buf = C.create_string_buffer(512)
        self._ck(self.lib.mm_comm_info(self.h, C.byref(n), buf, 512), "mm_comm_info")
        return {"world_seen": int(n.value), "library_path": buf.value.decode(errors="replace")}

    def allgatherv_mappings(self):
        self._ck(self.lib.mm_allgatherv_mappings(self.h), "mm_allgatherv_mappings")

    def allgatherv_mappings_begin(self):
        """start the exchange of the resident candidate mappings; the next batch may be mapped before allgatherv_mappings_end()"""
        self._ck(self.lib.mm_allgatherv_mappings_begin(self.h), "mm_allgatherv_mappings_begin")

    def allgatherv_mappings_end(self):
        self._ck(self.lib.mm_allgatherv_mappings_end(self.h), "mm_allgatherv_mappings_end")

    def gathered(self, world):
        counts = np.zeros(world, dtype=np.uint64); tot = C.c_size_t()
        self._ck(self.lib.mm_gathered_counts(self.h, _ptr(counts), C.byref(tot)), "mm_gathered_counts")
        out = np.zeros(tot.value, dtype=MAPPING_DT)
        self._ck(self.lib.mm_gathered_download(self.h, _ptr(out), tot.value), "mm_gathered_download")
        return out, counts.astype(np.int64)

    def index_replicate_from(self, src):
        self._ck(self.lib.mm_index_replicate(self.h, src.h), "mm_index_replicate")

    def bench_hash_only(self, reps=3):
        ms = C.c_double()
        self._ck(self.lib.mm_bench_hash_only(self.h, reps, C.byref(ms)), "mm_bench_hash_only")
        return ms.value

    def synchronize(self):
        self._ck(self.lib.mm_synchronize(self.h), "mm_synchronize")
