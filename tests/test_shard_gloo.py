"""The N>1 data path on CPU: world_size-2 (and 3) gloo processes shard a batch of reads the way skch::Map / bench.py do, hold
rank-local candidate mappings (mm_mapping records), exchange them with the product's all-gatherv protocol -- slots from
mashmap_amd/csrc/mm_exchange_plan.h, the header mm_comm.hip places its RCCL broadcasts by -- and run the product's host stage
(MapPost through libmashmap_host.so: chaining + plane-sweep filter) on the gathered records.  Every rank must end up with the
rows a single process computes from all records.  No GPU, no kernels: the records are synthetic; what is checked is the sharding
arithmetic, the exchange layout and that the CPU filters see input order."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mashmap_amd import shard  # noqa: E402

K, SEG, SKETCH, PI = 19, 5000, 130, 0.85
CONTIGS = np.array([4_000_000, 2_500_000, 1_000_000], dtype=np.int32)
ROW_DT = np.dtype([("querySeqId", "<i4"), ("queryLen", "<i4"), ("queryStartPos", "<i4"), ("queryEndPos", "<i4"), ("refSeqId", "<i4"), ("refStartPos", "<i4"),
                   ("refEndPos", "<i4"), ("strand", "<i4"), ("conservedSketches", "<i4"), ("blockLength", "<i4"), ("nucIdentity", "<f4"), ("kmerComplexity", "<f4")])


def test_read_blocks_partition_the_input():
    for n in (0, 1, 7, 1000, 1001):
        for world in (1, 2, 3, 8):
            blocks = [shard.read_block(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    lens = np.array([10, 10000, 5, 5, 20000, 7, 30000, 1, 1, 1])
    for world in (1, 2, 4):
        blocks = [shard.read_block_by_bases(lens, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == len(lens)
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))


def test_exchange_plan_is_rank_major_without_padding():
    assert shard.exchange_plan([3, 0, 5]).tolist() == [0, 3, 3, 8]
    assert shard.exchange_plan([0]).tolist() == [0, 0]
    assert shard.exchange_plan([7, 1]).tolist() == [0, 7, 8]


def _read_lens(n_reads):
    r = np.random.default_rng(n_reads)
    return r.choice([700, 5000, 10000, 12345, 23000], size=n_reads).astype(np.int32)


def _records(first, last, lens):
    """deterministic candidate mappings of the reads [first, last): what a rank's kernels would leave for its block.  Per fragment of
    a read 0..2 loci (a true one on a diagonal, sometimes a weaker decoy elsewhere), as mm_mapping records, read-major."""
    from mashmap_amd import capi
    out = []
    for read in range(first, last):
        L = int(lens[read])
        if L < K:
            continue
        frags = [(i * SEG, SEG) for i in range(L // SEG)] if L > SEG else [(0, L)]
        if L > SEG and L % SEG:
            frags.append((L - SEG, SEG))
        ctg = read % len(CONTIGS)
        origin = 10_000 + (read * 7919) % (int(CONTIGS[ctg]) - 60_000)
        strand = 1 if read % 3 else -1
        for fs, fl in frags:
            if (read + fs // SEG) % 11 == 5:
                continue                                               # an unmapped fragment
            pos = origin + (fs if strand > 0 else L - fs - fl)
            out.append((read, fs, fl, ctg, pos, 20 + (read + fs) % 60, SKETCH, strand, SKETCH, 0, 0x0100000000000000 + read))
            if (read + fs // SEG) % 4 == 1:                             # a decoy on another contig with fewer shared sketch elements
                out.append((read, fs, fl, (ctg + 1) % len(CONTIGS), 5_000 + (read * 104729) % 900_000, 9 + read % 5, SKETCH, -strand, SKETCH, 0, 0x0100000000000000 + read))
    return np.array(out, dtype=capi.MAPPING_DT) if out else np.zeros(0, dtype=capi.MAPPING_DT)


def _post(recs, lens, first_seq):
    lib = shard.host_lib()
    rows = np.zeros(len(recs) + 8, dtype=ROW_DT)
    rl = np.ascontiguousarray(lens, dtype=np.int32)
    recs = np.ascontiguousarray(recs)
    sec = C.c_double()
    n = lib.mmh_post_batch(K, SEG, SKETCH, PI, 1, 1, 1, len(CONTIGS), CONTIGS.ctypes.data, recs.ctypes.data, len(recs), rl.ctypes.data, len(rl), first_seq, 2,
                           C.byref(sec), rows.ctypes.data, len(rows))
    return rows[:n].copy()


def _worker(rank, world, port, n_reads, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = _read_lens(n_reads)
    a, b = shard.read_block_by_bases(lens, rank, world)               # the cut skch::Map::blocksOf makes
    local = _records(a, b, lens)
    gathered, counts = shard.allgatherv_mappings(local, dist)
    rows = _post(gathered, lens, 0)                                   # the CPU filters on ALL ranks' records, as rank 0 of a real run would
    dist.barrier()
    q.put((rank, counts, gathered.tobytes(), rows.tobytes()))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_reads,world", [(57, 2), (240, 2), (1, 2), (90, 3)])
def test_allgatherv_then_host_stage_world2_gloo(n_reads, world):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lens = _read_lens(n_reads)
    single = _records(0, n_reads, lens)                               # what one GPU would have produced for all reads
    rows1 = _post(single, lens, 0)
    assert n_reads < 5 or len(rows1) > 0
    for rank, counts, gathered, rows in res:
        assert sum(counts) == len(single)
        assert gathered == single.tobytes(), "rank-major gathered records are not the single-process records (rank %d)" % rank
        assert rows == rows1.tobytes(), "host stage on the gathered records differs from the single-process rows (rank %d)" % rank
