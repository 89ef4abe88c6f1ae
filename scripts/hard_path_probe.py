"""What does the sketch step cost when a share of the fragments is repeat-rich (tandem repeats, low complexity, N runs) and leaves the
fast kernel for the hard list?  Prints sketch / sketch_hard kernel ms per pass for several shares."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import bench as B
from mashmap_amd import capi
dev = torch.device('cuda', 0)
W = dict(B.WORKLOADS["configs1"])
NR = int(os.environ.get("READS", 400000)); L = W["read_len"]
contigs = B.make_reference(torch, dev, 4, 10_000_000)
ref_np = [c.cpu().numpy() for c in contigs]
ctx = capi.Context(k=W["k"], segLength=W["seg"], sketchSize=W["sketch"], flags=capi.MM_FLAG_HG_FILTER, device=0)
ctx.index_build(ref_np, kmerPct=0.001); ctx.set_tables_default(W["pi"])
base = B.make_reads(torch, dev, contigs, NR, L, W["err"], seed=5)
g = torch.Generator(device=dev); g.manual_seed(3)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
for kind in ("tandem2k", "homopolymer1k", "nrun", "satellite_all"):
    for share in (0.0, 0.01, 0.05, 0.25):
        if share == 0.0 and kind != "tandem2k": continue
        reads = base.clone().view(NR, L)
        n = int(NR * share)
        if n:
            rows = torch.randperm(NR, generator=g, device=dev)[:n]
            if kind == "tandem2k":                                   # 2 kbp of a 50 bp unit in the first fragment
                unit = lut[torch.randint(0, 4, (n, 50), generator=g, device=dev)]
                reads[rows, 1000:3000] = unit.repeat(1, 40)
            elif kind == "homopolymer1k":
                reads[rows, 2000:3000] = ord('A')
            elif kind == "nrun":
                reads[rows, 500:2500] = ord('N')
            else:                                                     # the whole read is a 171 bp satellite
                unit = lut[torch.randint(0, 4, (n, 171), generator=g, device=dev)]
                reads[rows] = unit.repeat(1, L // 171 + 1)[:, :L]
        flat = reads.reshape(-1).contiguous()
        torch.cuda.synchronize()                                      # the library packs on its own stream
        ctx.reads_upload_device(flat.data_ptr(), flat.numel(), np.arange(NR + 1, dtype=np.int64) * L)
        ctx.map(); ctx.profile(True); ctx.profile_read(reset=True)
        for _ in range(3): ctx.map()
        ctx.synchronize()
        p = ctx.profile_read(reset=True); ctx.profile(False)
        ms = {k: v[0] / 3 for k, v in p.items()}
        print("%-14s share %.2f: sketch %.2f ms, sketch_hard %.2f ms, all kernels %.2f ms (%d fragments)" % (kind, share, ms.get("sketch", 0), ms.get("sketch_hard", 0), sum(ms.values()), NR * 2), flush=True)
