"""Read-block arithmetic of the multi-GPU layout: one process per GPU, reads sharded in contiguous blocks, index
replicated.  The exchange step of the product is the RCCL all-gatherv inside libmashmap_hip.so (mashmap_amd/csrc/mm_comm.hip:
mm_allgatherv_mappings), which bench.py and skch::Map call; `allgatherv_records` below is the same collective over torch.distributed,
kept for the CPU tests (gloo) that pin down the layout -- rank-major records, contiguous read blocks == input order -- where no GPU
and no RCCL exist.
"""
import numpy as np

L2_WORDS = 8          # int32 words per mm_l2_locus: frag, cand, seqId, meanOptimalPos, optimalStart, optimalEnd, sharedSketchSize, strand


def read_block(n_reads, rank, world):
    """contiguous block [start, end) of reads for `rank`: blocks differ by at most one read and concatenate, in rank order,
    to the input order -- which is what keeps the PAF order of a sharded run equal to the single-GPU one"""
    base, extra = divmod(n_reads, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def read_block_by_bases(read_lengths, rank, world):
    """same, balancing bases instead of read counts (reads of very different lengths): cut points at equal shares of the
    cumulative length; deterministic and identical on every rank"""
    lens = np.asarray(read_lengths, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = int(cum[-1])
    cuts = [int(np.searchsorted(cum, (total * r) // world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, len(lens)
    for i in range(1, world + 1):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def allgatherv_records(local, dist, device=None, words=L2_WORDS):
    """all-gatherv of an (n_local, words) int32 tensor; returns (gathered (N, words) tensor, counts list).
    Counts first (one all_gather of a scalar per rank), then one padded all_gather; rank r's records land at
    sum(counts[:r]) in the result."""
    import torch
    world = dist.get_world_size()
    device = device if device is not None else local.device
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    counts_t = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts_t, n)
    counts = [int(c.item()) for c in counts_t]
    mx = max(counts) if counts else 0
    if mx == 0:
        return torch.zeros((0, words), dtype=torch.int32, device=device), counts
    mine = torch.zeros((mx, words), dtype=torch.int32, device=device)
    mine[:local.shape[0]] = local
    if dist.get_backend() == "nccl":
        buf = torch.empty((world * mx, words), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(buf, mine)
        parts = [buf[r * mx:r * mx + counts[r]] for r in range(world)]
    else:
        bufs = [torch.empty((mx, words), dtype=torch.int32, device=device) for _ in range(world)]
        dist.all_gather(bufs, mine)
        parts = [bufs[r][:counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0), counts


def globalise_fragments(gathered, counts, frags_per_rank):
    """rewrite the rank-local fragment ids (column 0) of gathered records into global fragment ids: rank r's fragments
    follow those of ranks < r (reads are sharded in contiguous blocks, so this is the single-GPU numbering)"""
    import torch
    out = gathered.clone()
    off, base = 0, 0
    for r, c in enumerate(counts):
        out[off:off + c, 0] += base
        off += c
        base += int(frags_per_rank[r])
    return out
