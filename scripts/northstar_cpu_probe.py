#!/usr/bin/env python3
"""The north_star target workload (10 kbp ONT-like reads, pi 85, defaults, against the human-scale 3 Gbp reference of bench.py's
`northstar` workload) through BOTH command lines on the same FASTA files: mashmap_hip (GPU) and the stock binary built from the
reference sources (oracle/_ref/mashmap_ref, -t THREADS).  The stock binary derives sketchSize 310 for the 3 GB reference file by
itself (no -J).  Prints both programs' own timers, the CPU mapping rate on this box's host cores, and whether the PAF files are
byte-identical.  usage: northstar_cpu_probe.py [--reads N] [--threads T]   (the reads are a sample of the benchmark's: the CPU leg
indexes 3 Gbp once -- about a minute on 64 threads -- so the sample only has to be large enough to time the mapping phase)"""
import argparse, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=30000)
ap.add_argument("--threads", type=int, default=64)
ap.add_argument("--contigs", type=int, default=0)
args = ap.parse_args()
import torch
dev = torch.device("cuda", 0)
W = dict(B.WORKLOADS["northstar"])
if args.contigs: W["ref_contigs"] = args.contigs
contigs = B.make_reference(torch, dev, W["ref_contigs"], W["ref_contig_len"])
td = tempfile.mkdtemp(prefix="mm_ns_")
rp, qp = os.path.join(td, "ref.fa"), os.path.join(td, "reads.fa")
t0 = time.time()
B.write_fasta(rp, ["chr%d" % i for i in range(len(contigs))], [c.cpu().numpy() for c in contigs])
L = W["read_len"]
rd = B.make_reads(torch, dev, contigs, args.reads, L, W["err"], seed=1000).cpu().numpy().reshape(args.reads, L)
B.write_fasta(qp, ["read%d" % i for i in range(args.reads)], list(rd), width=L)
with open(qp + ".fai", "w") as f:
    off = 0
    for i in range(args.reads):
        hdr = len(">read%d\n" % i); off += hdr
        f.write("read%d\t%d\t%d\t%d\t%d\n" % (i, L, off, L, L + 1)); off += L + 1
del contigs
torch.cuda.empty_cache()
print("north_star probe: %d x %d bp reads vs %.0f Mbp (%d contigs), FASTA written in %.0f s; mashmap -r ref.fa -q reads.fa -t %d (defaults: pi 85, segLength 5000)"
      % (args.reads, L, W["ref_contigs"] * W["ref_contig_len"] / 1e6, W["ref_contigs"], time.time() - t0, args.threads), flush=True)
out, times = {}, {}
for name, exe in (("hip", os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")), ("ref", os.path.join(ROOT, "oracle", "_ref", "mashmap_ref"))):
    if not os.path.exists(exe):
        continue
    t0 = time.time()
    p = subprocess.run([exe, "-r", rp, "-q", qp, "-o", os.path.join(td, name + ".paf"), "-t", str(args.threads)], capture_output=True, text=True)
    wall = time.time() - t0
    tm = {k: float(v) for k, v in re.findall(r"time spent (computing the reference index|mapping the query)\s*:\s*([0-9.eE+-]+)", p.stderr)}
    sk = re.findall(r"[Ss]ketch size\s*=?\s*(\d+)", p.stderr)
    print(name, "rc", p.returncode, "wall %.1f s" % wall, tm, "sketch size", sk[:1], flush=True)
    if p.returncode:
        print(p.stderr[-1500:])
    out[name] = open(os.path.join(td, name + ".paf"), "rb").read() if p.returncode == 0 else b""
    times[name] = tm
bases = args.reads * L
for name in times:
    m = times[name].get("mapping the query")
    if m:
        print("%s: mapping phase %.3f s = %.3f Gbp/s (FASTA -> PAF, its own timer)%s" % (name, m, bases / m / 1e9, " on %d host threads" % args.threads if name == "ref" else ""))
print("lines", {k: v.count(b"\n") for k, v in out.items()})
if len(out) == 2:
    print("PAF identical:", out["hip"] == out["ref"] and len(out["hip"]) > 0)
