"""mm_pack_read (mashmap_amd/host/pack2bit.hpp: makeUpperCaseAndValidDNA, commonFunc.hpp:97, + 2-bit packing on the host) against a
plain restatement of k_pack2bit's per-base rule: every byte value, every tail length, the AVX-512 / AVX2 / portable paths (the widest
one the CPU has by default; MASHMAP_HIP_PACK_ISA narrows it, one subprocess per setting).  CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

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


def check_every_byte_value_and_tail():
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


def test_pack_read_every_byte_value_and_tail():
    check_every_byte_value_and_tail()


@pytest.mark.parametrize("isa", ["scalar", "avx2", "avx512"])
def test_pack_read_with_the_instruction_set_narrowed(isa):
    """the dispatcher reads MASHMAP_HIP_PACK_ISA once per process: a fresh interpreter per setting (a CPU without the set falls back)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASHMAP_HIP_PACK_ISA=isa, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-c", "import tests.test_pack2bit as t; t.check_every_byte_value_and_tail(); print('ok')"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stderr[-2000:]
