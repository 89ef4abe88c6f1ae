"""Do two contexts on ONE GPU (each with half of the batch, each on its own stream and host thread) finish a batch sooner than one
context with the whole batch?  The sketch kernel is VALU-bound, lookup / locate / sweep wait on memory: kernels of the two kinds
running side by side could share a CU.  Prints ms per (whole) batch for 1 context and for 2 contexts, phase-shifted or not."""
import os, sys, time, threading
sys.path.insert(0, '.')
import numpy as np, torch
import bench as B
from mashmap_amd import capi
dev = torch.device('cuda', 0)
W = dict(B.WORKLOADS[os.environ.get("WL", "configs1")])
NR = int(os.environ.get("READS", W["reads"]))
STEPS = int(os.environ.get("STEPS", 6))
contigs = B.make_reference(torch, dev, W["ref_contigs"], W["ref_contig_len"])
ref_np = [c.cpu().numpy() for c in contigs]
reads = B.make_reads(torch, dev, contigs, NR, W["read_len"], W["err"], seed=1000)
torch.cuda.synchronize(); del contigs


def make_ctx(src=None):
    c = capi.Context(k=W["k"], segLength=W["seg"], sketchSize=W["sketch"], flags=capi.MM_FLAG_HG_FILTER, device=0)
    if src is None: c.index_build(ref_np, kmerPct=0.001)
    else: c.index_replicate_from(src)
    c.set_tables_default(W["pi"])
    return c


def upload(c, lo, hi):
    L = W["read_len"]
    t = reads[lo * L:hi * L]
    c.reads_upload_device(t.data_ptr(), t.numel(), np.arange(hi - lo + 1, dtype=np.int64) * L, seqCounterBase=lo)


a = make_ctx(); b = make_ctx(a)
upload(a, 0, NR)
a.map(); a.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS): a.map()
a.synchronize()
one = (time.perf_counter() - t0) / STEPS * 1e3
print("1 context, %d reads: %.2f ms / batch" % (NR, one))
n1 = a.result_counts()

for shift in (False, True):
    upload(a, 0, NR // 2); upload(b, NR // 2, NR)
    a.map(); b.map()
    def loop(c, delay):
        if delay: time.sleep(delay)
        for _ in range(STEPS): c.map()
        c.synchronize()
    ta = threading.Thread(target=loop, args=(a, 0)); tb = threading.Thread(target=loop, args=(b, one * 0.25e-3 if shift else 0))
    t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join()
    two = (time.perf_counter() - t0) / STEPS * 1e3
    na, nb = a.result_counts(), b.result_counts()
    print("2 contexts x %d reads, %s: %.2f ms / batch  (x%.3f)  L1 %d+%d vs %d" % (NR // 2, "phase-shifted" if shift else "in step", two, one / two, na[0], nb[0], n1[0]))
