"""Host-side plumbing of the multi-GPU layout: one process per GPU, reads sharded in contiguous blocks, index
replicated, ONE exchange step -- the all-gatherv of the candidate mappings (mm_mapping, 48 bytes) before the CPU filters.

In the product that exchange is RCCL inside libmashmap_hip.so (mashmap_amd/csrc/mm_comm.hip: a count all-gather, then `world`
broadcasts, root r into slot r of the gathered buffer).  `allgatherv_mappings` below runs the SAME protocol over a torch.distributed
group -- slots placed by the product's own mm_exchange_plan.h (through libmashmap_host.so: mmh_exchange_plan) -- so that the layout
contract (rank-major records == input order, no padding, empty slots skipped alike by every rank) is exercised by world-size-2 gloo
processes where no GPU and no RCCL exist (tests/test_shard_gloo.py, bench.py --stub)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(HERE, "lib", "libmashmap_host.so")
_host = None


def host_lib():
    """libmashmap_host.so: the CPU half of the product (exchange plan, MapPost = chaining + filters); no GPU needed"""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("%s not built (make -C mashmap_amd/host)" % HOST_LIB)
        lib = C.CDLL(HOST_LIB)
        lib.mmh_exchange_plan.restype = C.c_uint64
        lib.mmh_exchange_plan.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.mmh_post_batch.restype = C.c_int64
        lib.mmh_post_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_size_t, C.c_int32, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_size_t]
        _host = lib
    return _host


def exchange_plan(counts):
    """slot displacements (records) of the gathered buffer from the per-rank record counts: disp[r] .. disp[r+1] is rank r's slot"""
    cnt = np.ascontiguousarray(counts, dtype=np.uint64)
    disp = np.zeros(len(cnt) + 1, dtype=np.uint64)
    total = host_lib().mmh_exchange_plan(cnt.ctypes.data, len(cnt), disp.ctypes.data)
    assert int(disp[-1]) == int(total)
    return disp


def read_block(n_reads, rank, world):
    """contiguous block [start, end) of reads for `rank`: blocks differ by at most one read and concatenate, in rank order,
    to the input order -- which is what keeps the PAF order of a sharded run equal to the single-GPU one"""
    base, extra = divmod(n_reads, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def read_block_by_bases(read_lengths, rank, world):
    """same, balancing bases instead of read counts (reads of very different lengths): cut points at equal shares of the
    cumulative length; deterministic and identical on every rank (skch::Map::blocksOf cuts a batch the same way)"""
    lens = np.asarray(read_lengths, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = int(cum[-1])
    cuts = [int(np.searchsorted(cum, (total * r) // world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, len(lens)
    for i in range(1, world + 1):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def allgatherv_mappings(local, dist, device=None):
    """the exchange step over a torch.distributed group.  local: this rank's candidate mappings, a numpy structured array of 48-byte
    records (capi.MAPPING_DT).  Returns (all ranks' records, rank-major; counts).  Protocol of mm_comm.hip::exchange: counts first
    (one all_gather of a scalar), slots from mm_exchange_plan.h, then one broadcast per non-empty slot, root r -> slot r."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    rec = local.dtype.itemsize
    n = torch.tensor([len(local)], dtype=torch.int64, device=device)
    counts_t = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts_t, n)
    counts = [int(c.item()) for c in counts_t]
    disp = exchange_plan(counts)
    buf = torch.zeros(int(disp[-1]) * rec, dtype=torch.uint8, device=device)
    mine = torch.from_numpy(np.frombuffer(local.tobytes(), dtype=np.uint8).copy()).to(buf.device)
    for r in range(world):
        if counts[r] == 0:
            continue
        slot = buf[int(disp[r]) * rec:int(disp[r + 1]) * rec]
        if r == rank:
            slot.copy_(mine)
        dist.broadcast(slot, src=r)
    return np.frombuffer(buf.cpu().numpy().tobytes(), dtype=local.dtype), counts
