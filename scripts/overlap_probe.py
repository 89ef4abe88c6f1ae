"""Does running two contexts on one GPU (sketch of one overlapping the latency-bound map kernels of the other) help?"""
import os, sys, time, threading
sys.path.insert(0, '.')
import numpy as np, torch
import bench as B
from mashmap_amd import capi
dev = torch.device('cuda', 0)
N = int(os.environ.get("N", 1000000))
contigs = B.make_reference(torch, dev, B.REF_CONTIGS, B.REF_CONTIG_LEN)
ref_np = [c.cpu().numpy() for c in contigs]
reads = B.make_reads(torch, dev, contigs, N, B.READ_LEN, B.ERR, seed=1000)
def mk(lo, hi):
    ctx = capi.Context(k=B.K, segLength=B.SEG, sketchSize=B.SKETCH, flags=capi.MM_FLAG_HG_FILTER)
    ctx.index_build(ref_np, kmerPct=0.001); ctx.set_tables_default(B.PI)
    sub = reads[lo * B.READ_LEN:hi * B.READ_LEN]
    ctx.reads_upload_device(sub.data_ptr(), sub.numel(), np.arange(hi - lo + 1, dtype=np.int64) * B.READ_LEN)
    return ctx
one = mk(0, N)
one.map(); torch.cuda.synchronize()
t0 = time.perf_counter(); one.map(); one.map(); one.synchronize(); t1 = (time.perf_counter() - t0) / 2
print("one context, %d reads: %.1f ms/step" % (N, t1 * 1e3))
one.close()
for parts in (2, 4):
    ctxs = [mk(N * i // parts, N * (i + 1) // parts) for i in range(parts)]
    def run(c, reps):
        for _ in range(reps): c.map()
        c.synchronize()
    ths = [threading.Thread(target=run, args=(c, 1)) for c in ctxs]
    [t.start() for t in ths]; [t.join() for t in ths]
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(c, 2)) for c in ctxs]
    [t.start() for t in ths]; [t.join() for t in ths]
    t2 = (time.perf_counter() - t0) / 2
    print("%d contexts x %d reads concurrently: %.1f ms/step -> %.1f Gbp/s" % (parts, N // parts, t2 * 1e3, N * B.READ_LEN / t2 / 1e9))
    [c.close() for c in ctxs]
