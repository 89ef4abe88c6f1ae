"""The rule of the device's interval-point pre-filter (mashmap_amd/csrc/mm_map.hip: k_filter_points), as a Python model -- test infrastructure.

A fragment whose interval points do not fit the fused lookup kernel (repeat-rich references: hundreds of scattered hits per fragment) has them
gathered to HBM, sorted there and swept by the L1 kernels.  Most of those points are intervals that can never be part of a position reaching
minimumHits overlapping intervals (computeL1CandidateRegions, computeMap.hpp:916-1116, windowLen == 0): dropping them BEFORE the sort leaves
the L1 candidates exactly as they are and the sort a tenth of its work.  An interval [o, c) of contig q is kept iff

  * some bin (q, p >> BIN_SHIFT) it intersects, p in [o, c), is intersected by at least minimumHits intervals -- the count of a position is at
    most the count of its bin, so an interval that fails this covers no position that reaches minimumHits (the device counts bins in a hashed
    table: collisions only raise counts, i.e. keep more) --, or
  * it opens at the smallest position or closes at the largest position of its contig among the fragment's points: the reference's sweep
    groups points by `pos` alone (computeMap.hpp:967, :1047-1051), so the last point of one contig and the first of the next may share a
    group; with both boundary groups kept as they are, that seam behaves as it did.

tests/test_l1_point_filter.py checks the rule against the literal L1 of the oracle on fuzzed point sets."""
import numpy as np

BIN_SHIFT = 12


def keep_mask(seq, o, c, min_hits, table_slots=None):
    """seq, o, c: arrays describing the intervals; returns the boolean keep mask.  table_slots: None = exact bins, else the size of the hashed
    counter table of the device (a power of two)"""
    seq = np.asarray(seq, dtype=np.int64); o = np.asarray(o, dtype=np.int64); c = np.asarray(c, dtype=np.int64)
    n = len(seq)
    if min_hits <= 1 or n == 0:
        return np.ones(n, dtype=bool)
    counts = {}

    def slot(q, b):
        if table_slots is None:
            return (int(q), int(b))
        return int(((int(q) * 0x9E3779B1 + int(b) * 0x85EBCA77) & 0xFFFFFFFF) >> 7) & (table_slots - 1)      # bin_slot() of mm_map.hip
    for i in range(n):
        for b in range(int(o[i]) >> BIN_SHIFT, ((int(c[i]) - 1) >> BIN_SHIFT) + 1):
            counts[slot(seq[i], b)] = counts.get(slot(seq[i], b), 0) + 1
    keep = np.zeros(n, dtype=bool)
    for i in range(n):
        keep[i] = any(counts[slot(seq[i], b)] >= min_hits for b in range(int(o[i]) >> BIN_SHIFT, ((int(c[i]) - 1) >> BIN_SHIFT) + 1))
    for q in np.unique(seq):
        m = seq == q
        keep |= m & (o == o[m].min())
        keep |= m & (c == c[m].max())
    return keep
