"""Mapping cost against a reference in which every locus has COPIES near-identical copies (segmental-duplication-like): per fragment
COPIES x the interval points and L1 candidates.  Prints kernel ms per pass and the record counts."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import bench as B
from mashmap_amd import capi
dev = torch.device('cuda', 0)
W = dict(B.WORKLOADS["configs1"])
NR = int(os.environ.get("READS", 200000)); L = W["read_len"]
for COPIES in [int(x) for x in os.environ.get("COPIES", "1,4,16").split(",")]:
    g = torch.Generator(device=dev); g.manual_seed(11)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    unit = lut[torch.randint(0, 4, (40_000_000 // COPIES,), generator=g, device=dev)]
    contigs = []
    for c in range(COPIES):
        m = torch.rand(unit.numel(), generator=g, device=dev) < 0.01
        contigs.append(torch.where(m, lut[torch.randint(0, 4, (unit.numel(),), generator=g, device=dev)], unit))
    ref_np = [c.cpu().numpy() for c in contigs]
    reads = B.make_reads(torch, dev, contigs, NR, L, W["err"], seed=5)
    torch.cuda.synchronize()
    ctx = capi.Context(k=W["k"], segLength=W["seg"], sketchSize=W["sketch"], flags=capi.MM_FLAG_HG_FILTER, device=0)
    ctx.index_build(ref_np, kmerPct=0.001); ctx.set_tables_default(W["pi"])
    ctx.reads_upload_device(reads.data_ptr(), reads.numel(), np.arange(NR + 1, dtype=np.int64) * L)
    ctx.map(); ctx.profile(True); ctx.profile_read(reset=True)
    for _ in range(2): ctx.map()
    ctx.synchronize()
    p = ctx.profile_read(reset=True); ctx.profile(False)
    ms = {k: round(v[0] / 2, 2) for k, v in p.items() if v[0] > 0.005}
    n1, n2 = ctx.result_counts()
    print("copies %2d: %d fragments, %d L1 candidates, %d L2 loci, %d candidate mappings; kernels %.1f ms: %s" % (COPIES, NR * 2, n1, n2, len(ctx.mappings()), sum(ms.values()), ms), flush=True)
    ctx.close(); del reads, contigs
    torch.cuda.empty_cache()
