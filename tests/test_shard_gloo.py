"""The N>1 data path on CPU: world_size-2 gloo processes shard reads, produce rank-local L2 records and exchange them with
the same all-gatherv bench.py uses over RCCL.  No GPU, no kernels: this checks the sharding arithmetic and the collective."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mashmap_amd import shard  # noqa: E402


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


def _fake_records(rank_start, rank_end, frags_per_read=2):
    """deterministic stand-in for a rank's L2 output: one record per fragment, some fragments unmapped"""
    recs = []
    for read in range(rank_start, rank_end):
        for j in range(frags_per_read):
            gfrag = read * frags_per_read + j
            if gfrag % 5 == 3:
                continue
            local = (read - rank_start) * frags_per_read + j
            recs.append([local, local, gfrag % 7, 1000 + gfrag, 900 + gfrag, 1100 + gfrag, 50 + gfrag % 13, 1 if gfrag % 2 else -1])
    return np.array(recs, dtype=np.int32).reshape(-1, shard.L2_WORDS)


def _worker(rank, world, port, n_reads, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard.read_block(n_reads, rank, world)
    local = torch.from_numpy(_fake_records(a, b))
    gathered, counts = shard.allgatherv_records(local, dist)
    frags = [2 * (shard.read_block(n_reads, r, world)[1] - shard.read_block(n_reads, r, world)[0]) for r in range(world)]
    glob = shard.globalise_fragments(gathered, counts, frags)
    dist.barrier()
    q.put((rank, counts, glob.numpy().copy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_reads", [11, 40, 1])
def test_allgatherv_world2_gloo(n_reads):
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _fake_records(0, n_reads)                      # what one GPU would have produced for all reads
    for rank, counts, glob in res:
        assert sum(counts) == len(single)
        assert np.array_equal(glob[:, 0], single[:, 0])    # global fragment ids == single-GPU numbering
        assert np.array_equal(glob[:, 2:], single[:, 2:])
