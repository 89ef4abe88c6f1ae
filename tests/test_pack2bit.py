"""mm_pack_read (mashmap_amd/host/pack2bit.hpp: makeUpperCaseAndValidDNA, commonFunc.hpp:97, + 2-bit packing on the host) against a
plain restatement of k_pack2bit's per-base rule: every byte value, every tail length, AVX2 and portable paths.  CPU only."""
import numpy as np

from mashmap_amd import capi


def ref_pack(a):
    n = len(a); g = (n + 31) // 32
    b2 = np.zeros(2 * g, dtype=np.uint32); nm = np.zeros(g, dtype=np.uint32)
    for i, ch in enumerate(a):
        c = int(ch) & 0xDF                                        # a-z -> A-Z
        ok = c in (65, 67, 71, 84)
        code = ((c >> 1) ^ (c >> 2)) & 3 if ok else 0             # A0 C1 G2 T3, N -> 0
        b2[i // 16] |= np.uint32(code << (2 * (i % 16)))
        if not ok:
            nm[i // 32] |= np.uint32(1 << (i % 32))
    return b2, nm


def test_pack_read_every_byte_value_and_tail():
    rng = np.random.default_rng(1)
    reads = [np.arange(256, dtype=np.uint8), rng.integers(0, 256, 1000).astype(np.uint8), np.frombuffer(b"ACGTacgtNnRYKM-*xX", dtype=np.uint8)]
    reads += [rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), n) for n in (0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 97, 5000)]
    for portable in (False, True):
        b2, nm, hasn, lens = capi.pack_reads(reads, portable)
        at = 0
        for i, r in enumerate(reads):
            e2, en = ref_pack(r); g = len(en)
            assert (b2[2 * at:2 * at + 2 * g] == e2).all() and (nm[at:at + g] == en).all(), (portable, i)
            assert hasn[i] == (1 if en.any() else 0) and lens[i] == len(r)
            at += g
        assert at == len(nm)
