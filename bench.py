#!/usr/bin/env python3
"""bench.py -- query Gbp/s of the sketch + L1/L2 hot path on N MI355X (one process per GPU).

  python bench.py --gpus N --steps K --warmup W [--workload configs1|configs3|configs4|northstar] [--batches B]

With N > 1 and no torch.distributed environment the script starts its own N ranks (torch.distributed.run, 127.0.0.1); started by
a launcher it checks that WORLD_SIZE == N.  It never prints an `n_gpus` other than the N it was asked for.

A "step" = one pass of the hot path (sketch -> seed lookup -> L1 -> L2 slide -> doL2Mapping's best-first selection) over one resident
batch; inputs (2-bit packed bases + N mask) are already in HBM when the timed region starts.  B = 3 DISTINCT batches of the workload
(different reads, same shape) are resident and take turns (mm_reads_exchange: a pointer swap), so a pass is never sized by a previous
pass over the same reads; `passes` in the line says how many of the timed passes went through as steady-state passes (one host wait)
and how many outgrew a buffer and were redone.  For N > 1 every rank maps its own reads (weak scaling: `reads` per GPU and batch, index
replicated) and the step ends with the RCCL all-gatherv of the candidate mappings (mm_allgatherv_mappings_begin/_end,
mashmap_amd/csrc/mm_comm.hip) -- the product's own exchange step, overlapped with the next batch, the last one waited for inside the
timed region.

Workloads (BASELINE.json `configs`; the default is configs[1], the configuration the metric is quoted on):
  configs1  1 M x 10 kbp reads (10 % ONT-like error) vs 100 Mbp, pi 85, segLength 5000, sketchSize 130
  configs3  per-GPU share of configs[3]: 1.25 M x 15 kbp reads vs 3 Gbp (24 x 125 Mbp), sketchSize 310 (the stock binary's value:
            its int32 referenceSize overflows for a 3 GB file; 220 mathematically -- SURVEY App. C)
  northstar the north_star target sentence: 1 M x 10 kbp reads, pi 85, against the 3 Gbp index (sketchSize 310 as for configs3)
  configs4  per-GPU share of configs[4]: 625 k x 20 kbp reads at 15-20 % error vs 10 x 300 Mbp (the --rl list shares one seqId
            space, winSketch.hpp:174-214), --dense --pi 80 => sketchSize 498
--reads / --ref-contigs / --ref-contig-len scale a workload down; the JSON line names what actually ran.

The default run (configs1, nothing scaled, one GPU) carries three more measurements behind the headline one, as extra keys; `value` /
`config` / `roofline` stay the configs[1] figures:
  e2e                 the `mashmap_hip` command line FASTA -> PAF on configs[1] (10 GB of FASTA written to a temporary directory), per stage
  north_star_target   the north_star sentence -- 1 M x 10 kbp reads at pi 85 against the human-scale (3 Gbp) index --, stock segLength 5000
                      and the "10 kbp segments" variant, the stock binary beside it; and `repeat_rich`: the same workload on a reference
                      with human-like repeat structure (make_repeat_rich_reference: ~45 % interspersed repeat families, satellites, N gaps)
--no-e2e / --no-north-star skip them.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md)

WORKLOADS = {
    "configs1": dict(label="configs[1]", k=19, seg=5000, sketch=130, pi=0.85, read_len=10000, err=(0.10, 0.10), reads=1_000_000,
                     ref_contigs=10, ref_contig_len=10_000_000,
                     sketch_note="130 = recommendedSketchSize for a 100 Mbp reference file (SURVEY App. C)"),
    "configs2": dict(label="configs[2]", k=19, seg=10000, sketch=40, pi=0.95, read_len=125_000_000, err=(0.01, 0.01), reads=24,
                     ref_contigs=24, ref_contig_len=125_000_000, assembly=True, cli=["-f", "one-to-one"],
                     sketch_note="40 = what the stock binary derives at pi 95, segLength 10000 for a 3 GB reference file (int32 referenceSize overflow); 20 mathematically (SURVEY App. C); pinned with -J 40"),
    "configs3": dict(label="configs[3] (per-GPU share of 10 M reads / 8 GPUs)", k=19, seg=5000, sketch=310, pi=0.85, read_len=15000,
                     err=(0.10, 0.10), reads=1_250_000, ref_contigs=24, ref_contig_len=125_000_000,
                     sketch_note="310 = what the stock binary derives for a 3 GB reference file (int32 referenceSize overflow); 220 mathematically (SURVEY App. C)"),
    "northstar": dict(label="north_star target (10 kbp reads, pi 85, human-scale index)", k=19, seg=5000, sketch=310, pi=0.85, read_len=10000,
                      err=(0.10, 0.10), reads=1_000_000, ref_contigs=24, ref_contig_len=125_000_000,
                      sketch_note="310 = what the stock binary derives for a 3 GB reference file (int32 referenceSize overflow); 220 mathematically (SURVEY App. C)"),
    "configs4": dict(label="configs[4] (per-GPU share of 5 M reads / 8 GPUs)", k=19, seg=5000, sketch=498, pi=0.80, read_len=20000,
                     err=(0.15, 0.20), reads=625_000, ref_contigs=10, ref_contig_len=300_000_000,
                     sketch_note="498 = --dense at pi 80: 0.02 (1 + 0.2 / 0.05) (5000 - 19) (parseCmdArgs.hpp:620-641); the 10 --rl files are 10 contigs of one index"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_reference(torch, dev, ncontigs, clen, seed=1):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    out = []
    for _ in range(ncontigs):
        out.append(lut[torch.randint(0, 4, (clen,), generator=g, device=dev, dtype=torch.int32).long()] if clen <= (1 << 27)
                   else torch.cat([lut[torch.randint(0, 4, (min(1 << 27, clen - o),), generator=g, device=dev, dtype=torch.int32).long()]
                                   for o in range(0, clen, 1 << 27)]))
    return out


# Human-like repeat structure for `north_star_target.repeat_rich` and tests/humanscale.py (sizes are for a 3 Gbp reference; copy numbers scale
# with the reference so that the covered fraction stays): interspersed repeat families over ~45 % of the sequence -- copy numbers from 10^2
# to 10^5, copies diverged from their family's consensus by 10-20 % (i.i.d. substitutions, random strand), the numerous families the more
# diverged ones as in real genomes (old families are both) --, one satellite array per contig (171 bp monomers in a 12-monomer higher-order
# repeat, copies 2 % apart) and N gaps.  (family, families, consensus bp, copies per family at 3 Gbp, 5'-truncated copies)
REPEAT_FAMILIES = [("SINE-like", 10, 300, 100_000, False),               # 300 Mbp
                   ("LINE-like", 20, 6000, 10_000, True),                # copies keep the last 500..6000 bp: 650 Mbp
                   ("LTR/DNA-like", 100, 2000, 1_000, False),            # 200 Mbp
                   ("segmental-duplication-like", 200, 10_000, 100, False)]   # 200 Mbp
SATELLITE_BP, SATELLITE_MONOMER, SATELLITE_HOR, SATELLITE_DIV = 250_000, 171, 12, 0.02
NGAP_BP, NGAP_END_BP = 500_000, 10_000


def repeat_divergence(copies_at_3gbp):
    return 0.10 + 0.10 * (np.log10(copies_at_3gbp) - 2.0) / 3.0


def make_repeat_rich_reference(torch, dev, ncontigs, clen, seed=11):
    """a reference with the repeat structure described at REPEAT_FAMILIES, as `ncontigs` consecutive views of one uint8 tensor (ASCII);
    returns (contigs, summary)"""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    total = ncontigs * clen
    scale = total / 3e9
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    whole = torch.empty(total, dtype=torch.uint8, device=dev)
    for o in range(0, total, 1 << 27):
        n = min(1 << 27, total - o)
        whole[o:o + n] = torch.randint(0, 4, (n,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)      # codes 0..3 until the end
    covered = 0
    fams = []
    for name, nfam, clen_f, copies3, trunc in REPEAT_FAMILIES:
        copies = max(2, int(round(copies3 * scale)))
        div = float(repeat_divergence(copies3))
        ar = torch.arange(clen_f, device=dev)
        for _ in range(nfam):
            cons = torch.randint(0, 4, (clen_f,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
            for c0 in range(0, copies, 1 << 14):                      # 16 k copies at a time (a LINE-like block is 100 M cells)
                n = min(1 << 14, copies - c0)
                cp = cons[None, :].expand(n, clen_f).clone()
                sub = torch.rand(n, clen_f, generator=g, device=dev) < div
                cp = torch.where(sub, (cp + torch.randint(1, 4, (n, clen_f), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)) & 3, cp)
                keep = torch.ones(n, clen_f, dtype=torch.bool, device=dev)
                if trunc:
                    ln = torch.randint(min(500, clen_f), clen_f + 1, (n,), generator=g, device=dev)
                    keep = ar[None, :] >= (clen_f - ln)[:, None]
                rev = torch.rand(n, generator=g, device=dev) < 0.5
                cp = torch.where(rev[:, None], (3 - cp).flip(1), cp)
                keep = torch.where(rev[:, None], keep.flip(1), keep)
                ci = torch.randint(0, ncontigs, (n,), generator=g, device=dev)
                st = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * (clen - clen_f)).long()
                idx = (ci * clen + st)[:, None] + ar[None, :]
                whole[idx[keep]] = cp[keep]
                covered += int(keep.sum())
                del cp, sub, keep, idx
        fams.append({"family": name, "families": nfam, "consensus_bp": clen_f, "copies_per_family": copies, "divergence": round(div, 3)})
    # satellites: one array per contig at 40 % of its length
    sat_bp = min(SATELLITE_BP, clen // 20)
    hor_len = SATELLITE_MONOMER * SATELLITE_HOR
    for c in range(ncontigs):
        mono = torch.randint(0, 4, (SATELLITE_MONOMER,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
        hor = mono.repeat(SATELLITE_HOR)
        m = torch.rand(hor_len, generator=g, device=dev) < 0.25           # the monomers of the higher-order unit differ from each other
        hor = torch.where(m, (hor + torch.randint(1, 4, (hor_len,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)) & 3, hor)
        arr = hor.repeat(sat_bp // hor_len + 1)[:sat_bp]
        m = torch.rand(sat_bp, generator=g, device=dev) < SATELLITE_DIV
        arr = torch.where(m, (arr + torch.randint(1, 4, (sat_bp,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)) & 3, arr)
        o = c * clen + int(clen * 0.4)
        whole[o:o + sat_bp] = arr
    for o in range(0, total, 1 << 27):
        n = min(1 << 27, total - o)
        whole[o:o + n] = lut[whole[o:o + n].long()]
    gap, end = min(NGAP_BP, clen // 50), min(NGAP_END_BP, clen // 1000)
    for c in range(ncontigs):
        o = c * clen
        whole[o:o + end] = ord("N"); whole[o + clen - end:o + clen] = ord("N")
        whole[o + int(clen * 0.6):o + int(clen * 0.6) + gap] = ord("N")
    summary = {"generator": "bench.make_repeat_rich_reference(seed %d)" % seed, "interspersed_repeat_fraction": round(covered / total, 3), "families": fams,
               "satellite": "%d bp array per contig: %d bp monomers in a %d-monomer higher-order repeat, copies %.0f %% apart" % (sat_bp, SATELLITE_MONOMER, SATELLITE_HOR, SATELLITE_DIV * 100),
               "n_gaps": "%d bp inside every contig, %d bp at both ends" % (gap, end)}
    return [whole[c * clen:(c + 1) * clen] for c in range(ncontigs)], summary


def contiguous_views(torch, contigs):
    """the contigs on the host as consecutive views of ONE array: what a caller that has parsed its FASTA into one buffer hands to
    mm_index_build (capi.Context.index_build then passes the buffer as it lies instead of concatenating 3 GB inside the timed build)"""
    whole = torch.cat(contigs).cpu().numpy()
    out, at = [], 0
    for c in contigs:
        out.append(whole[at:at + len(c)]); at += len(c)
    return out


def make_reads(torch, dev, contigs, nreads, read_len, err, seed, chunk=8192):
    """ONT-like reads on the device: uniform start/strand, i.i.d. e/3 sub + e/3 ins + e/3 del with the read's error rate e drawn
    uniformly from err = (lo, hi)."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    ref = torch.cat(contigs)
    coff = torch.tensor(np.cumsum([0] + [len(c) for c in contigs[:-1]]), device=dev)
    clen = torch.tensor([len(c) for c in contigs], device=dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    src_len = int(read_len * (1 + err[1])) + 300
    out = torch.empty(nreads * read_len, dtype=torch.uint8, device=dev)
    ar = torch.arange(src_len, device=dev)
    for r0 in range(0, nreads, chunk):
        R = min(chunk, nreads - r0)
        ci = torch.randint(0, len(contigs), (R,), generator=g, device=dev)
        st = (torch.rand(R, generator=g, device=dev, dtype=torch.float64) * (clen[ci] - src_len).double()).long()
        rev = torch.rand(R, generator=g, device=dev) < 0.5
        e = (err[0] + (err[1] - err[0]) * torch.rand(R, generator=g, device=dev))[:, None]
        seg = ref[(coff[ci] + st)[:, None] + ar[None, :]]
        seg = torch.where(rev[:, None], comp[seg.flip(1).long()], seg)
        u = torch.rand(R, src_len, generator=g, device=dev)
        rb = lut[torch.randint(0, 4, (R, src_len), generator=g, device=dev)]
        is_sub = u < e / 3
        is_ins = (u >= e / 3) & (u < 2 * e / 3)
        is_del = (u >= 2 * e / 3) & (u < e)
        cnt = (~is_del).int() + is_ins.int()
        pos = torch.cumsum(cnt, dim=1) - cnt                     # output slot of the (possibly inserted) first symbol
        base = torch.where(is_sub & (rb != seg), rb, seg)
        dst = out[r0 * read_len:(r0 + R) * read_len].view(R, read_len)
        rows = torch.arange(R, device=dev)[:, None].expand(R, src_len)
        m = is_ins & (pos < read_len)
        dst[rows[m], pos[m]] = rb[m]
        p2 = pos + is_ins.int()
        m = (~is_del) & (p2 < read_len)
        dst[rows[m], p2[m]] = base[m]
        assert int((pos[:, -1] + cnt[:, -1]).min()) >= read_len
        del seg, u, rb, cnt, pos, base, rows, m, p2
    return out


def make_assembly(torch, dev, contigs, div=0.01, seed=21):
    """BASELINE configs[2]'s query: every reference contig with `div` i.i.d. substitutions and a few 1-5 Mbp
    rearrangements -- an inversion (contig i % 3 == 0), a translocation inside the contig (i % 3 == 1), both and the whole contig on the
    other strand (i % 3 == 2).  Lengths stay; returns one uint8 tensor per contig, on the device."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    rs = np.random.RandomState(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    comp = torch.arange(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    out = []
    for i, c in enumerate(contigs):
        n = len(c)
        q = c.clone()
        for o in range(0, n, 1 << 27):                         # substitutions, in pieces (the masks are 4 bytes per base)
            m = min(1 << 27, n - o)
            hit = torch.rand(m, generator=g, device=dev) < div * 4.0 / 3.0          # a drawn base equals the old one a quarter of the time
            q[o:o + m] = torch.where(hit, lut[torch.randint(0, 4, (m,), generator=g, device=dev)], q[o:o + m])
            del hit
        unit = max(1, min(1_000_000, n // 125))               # 1 Mbp at 125 Mbp contigs; scaled-down contigs keep the proportions
        if i % 3 in (0, 2) and n > 12 * unit:                 # inversion of 1..5 units
            ln = int(rs.randint(1, 6)) * unit; at = int(rs.randint(unit, n - ln - unit))
            q[at:at + ln] = comp[q[at:at + ln].flip(0).long()]
        if i % 3 in (1, 2) and n > 12 * unit:                 # translocation: a 1..5 unit piece moves towards the other end of the contig
            ln = int(rs.randint(1, 6)) * unit; at = int(rs.randint(unit, n // 2 - ln)); to = int(rs.randint(n // 2, n - unit))
            q = torch.cat([q[:at], q[at + ln:to], q[at:at + ln], q[to:]])
        if i % 3 == 2:
            q = comp[q.flip(0).long()]
        assert len(q) == n
        out.append(q)
    return out


def write_fasta(path, names, arrays, width=100):
    with open(path, "wb") as f:
        for n, a in zip(names, arrays):
            f.write(b">" + n.encode() + b"\n")
            full = (len(a) // width) * width
            if full:
                lines = np.concatenate([a[:full].reshape(-1, width), np.full((full // width, 1), 10, dtype=np.uint8)], axis=1)
                f.write(lines.tobytes())
            if len(a) > full:
                f.write(a[full:].tobytes() + b"\n")


def usable_cpus():
    """CPUs this process may use at once: hardware threads, affinity mask, and the container's CPU quota (cgroup v2 cpu.max / v1
    cfs_quota) -- the GPU boxes show 256 hardware threads and grant 16 CPUs' worth of time; more threads than that get the whole
    process throttled."""
    n = os.cpu_count() or 1
    if hasattr(os, "sched_getaffinity"):
        n = min(n, len(os.sched_getaffinity(0)) or n)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(W, ref_np, reads_np, n_sample):
    """the reference's own CPU path (oracle/_ref/mashmap_ref, built from /root/reference with the GSL stand-in) or, if that binary
    did not travel, our CPU port (oracle/liboracle.so); timed on this box's host cores on a bounded sample of the same workload."""
    ncores = os.cpu_count() or 1
    read_len = W["read_len"]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mashmap_ref")
    prof_bin = os.path.join(ROOT, "oracle", "_ref", "mashmap_ref_prof")
    sample = reads_np[:n_sample * read_len].reshape(n_sample, read_len)
    desc = "%d of the benchmark reads (%.0f Mbp) vs the same %.0f Mbp reference" % (n_sample, n_sample * read_len / 1e6, sum(len(a) for a in ref_np) / 1e6)
    if os.path.exists(ref_bin):
        with tempfile.TemporaryDirectory() as td:
            rp, qp, op = os.path.join(td, "ref.fa"), os.path.join(td, "q.fa"), os.path.join(td, "o.paf")
            write_fasta(rp, ["chr%d" % i for i in range(len(ref_np))], ref_np)
            write_fasta(qp, ["read%d" % i for i in range(n_sample)], list(sample))
            with open(qp + ".fai", "w") as f:          # avoids the reference's extra pass over the query file
                for i in range(n_sample):
                    f.write("read%d\t%d\t0\t100\t101\n" % (i, read_len))
            common = ["-r", rp, "-q", qp, "-o", op, "-s", str(W["seg"]), "--pi", str(int(round(W["pi"] * 100))), "-k", str(W["k"]), "-J", str(W["sketch"])]
            # the reference's pthread pool stops scaling early (one reader thread feeds it; with hundreds of threads it thrashes):
            # time a few thread counts on the same sample and report the best one
            best = None
            for nt in sorted({min(ncores, 8), min(ncores, 32), min(ncores, 64)}):
                t0 = time.time()
                p = subprocess.run([ref_bin] + common + ["-t", str(nt)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                wall = time.time() - t0
                tmap = None
                for line in p.stderr.splitlines():
                    if "time spent mapping the query" in line:
                        tmap = float(line.split(":")[-1].split()[0])
                if p.returncode == 0 and tmap:
                    log("[cpu_baseline] reference binary -t %d: map %.2f s (total wall %.1f s)" % (nt, tmap, wall))
                    if best is None or tmap < best[0]:
                        best = (tmap, nt)
            # sum of the per-fragment compute times of the -DENABLE_TIME_PROFILE_L1_L2 build (no reader, no
            # pool overhead) / threads = the rate an ideally fed pool of that many cores would reach
            compute = None
            if best and os.path.exists(prof_bin):
                p = subprocess.run([prof_bin] + common + ["-t", str(best[1])], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                # lines "seqCounter len tL1 tL2 tFragment" (computeMap.hpp:802-811); the pool's threads write them unsynchronised, so only
                # lines that parse cleanly are used and their mean is scaled to the number of fragments of the sample
                tot = 0.0; nfr = 0
                for line in p.stderr.splitlines():
                    f = line.split()
                    if len(f) == 5 and f[0].isdigit() and f[1] == str(W["seg"]):
                        try:
                            t = [float(x) for x in f[2:]]
                        except ValueError:
                            continue
                        if all(0 <= x < 10 for x in t) and abs(t[0] + t[1] - t[2]) < 1e-3:
                            tot += t[2]; nfr += 1
                if nfr:
                    tot = tot / nfr * (n_sample * (read_len // W["seg"] + (1 if read_len % W["seg"] else 0)))
                if p.returncode == 0 and nfr:
                    compute = {"what": "-DENABLE_TIME_PROFILE_L1_L2 build of the reference: per-fragment sketch+L1+L2 seconds, no reader, no pool overhead "
                                       "(SURVEY section 8d(b)); per core, and x host cores as the ideally fed pool", "fragments_parsed": nfr, "sum_fragment_seconds": round(tot, 3),
                               "gbps_per_core": round(n_sample * read_len / tot / 1e9, 5),
                               "gbps_all_cores_ideal": round(n_sample * read_len / tot / 1e9 * ncores, 3),
                               "gbps_usable_cpus_ideal": round(n_sample * read_len / tot / 1e9 * usable_cpus(), 3)}
            if best:
                tmap, nt = best
                return {"value": n_sample * read_len / tmap / 1e9, "unit": "Gbp/s", "cores": usable_cpus(), "threads": nt, "host_hardware_threads": ncores, "kind": "reference",
                        "sample": desc + "; mashmap_ref (built from the reference sources) best of -t 8/32/64 = -t %d; `cores` = the %d CPUs this process may use at once "
                                         "(affinity / container quota) of the host's %d hardware threads; 'time spent mapping the query' (includes its single-threaded FASTA reader)" % (nt, usable_cpus(), ncores),
                        "fragment_compute": compute}
            log("[cpu_baseline] reference binary failed, falling back to the port:", p.stderr[-300:])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mmutil as U
    orc = U.Oracle()
    h = orc.session([("chr%d" % i, a) for i, a in enumerate(ref_np[:1])], W["k"], W["seg"], W["sketch"], W["pi"])
    n = min(n_sample, 200)
    t0 = time.time()
    for i in range(n):
        for off in range(0, read_len - W["seg"] + 1, W["seg"]):
            orc.map_fragment(h, sample[i, off:off + W["seg"]], i, b"r", read_len, W["sketch"])
    dt = time.time() - t0
    orc.free(h)
    return {"value": n * read_len / dt / 1e9 * 0.5, "unit": "Gbp/s", "cores": 1, "kind": "port",
            "sample": "%d reads vs the first contig; scalar port, diagnostic entry runs the path twice (halved)" % n}


def host_path(ctx, W, nreads, ref_lens, steps_ms):
    """packed bases -> MappingResult rows: the device pass + download of the candidate mappings + the host stage of skch::Map
    (chaining, plane-sweep filter, sanity checks; libmashmap_host.so = MapPost) on every host core."""
    import ctypes as C
    from mashmap_amd import capi
    lib_path = os.path.join(ROOT, "mashmap_amd", "lib", "libmashmap_host.so")
    if not os.path.exists(lib_path):
        return None
    lib = C.CDLL(lib_path)
    lib.mmh_post_batch.restype = C.c_int64
    lib.mmh_post_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p, C.c_size_t, C.c_int32, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_size_t]
    t0 = time.perf_counter()
    recs = ctx.mappings()
    t_dl = time.perf_counter() - t0
    clens = np.ascontiguousarray(ref_lens, dtype=np.int32)
    rl = np.full(nreads, W["read_len"], dtype=np.int32)
    threads = os.cpu_count() or 1
    best = None
    for nt in sorted({min(threads, usable_cpus()), min(threads, 32), min(threads, 64), min(threads, 128), threads}):   # the quota-sized pool first: wider ones are throttled on a capped box
        sec = C.c_double()
        rows = lib.mmh_post_batch(W["k"], W["seg"], W["sketch"], W["pi"], 1, 1, 1, len(clens), clens.ctypes.data, recs.ctypes.data, len(recs),
                                  rl.ctypes.data, nreads, 0, nt, C.byref(sec), None, 0)
        if best is None or sec.value < best[0]:
            best = (sec.value, nt, int(rows))
    post_s, nt, rows = best
    bases = nreads * W["read_len"]
    dev_s = steps_ms / 1e3
    return {"what": "packed bases -> reported MappingResult rows on one GPU + host: device pass, D2H of the candidate mappings (48 B each), "
                    "then per read mergeMappingsInRange + filterByGroup + sanity checks (MapPost, the code skch::Map runs) on host threads",
            "candidate_mappings": int(len(recs)), "rows": rows, "device_ms": round(dev_s * 1e3, 3), "download_ms": round(t_dl * 1e3, 3),
            "host_ms": round(post_s * 1e3, 3), "host_threads": nt, "host_cores": threads, "usable_cpus": usable_cpus(),
            "gbps_serial": round(bases / (dev_s + t_dl + post_s) / 1e9, 3),
            "gbps_pipelined": round(bases / max(dev_s, t_dl + post_s) / 1e9, 3),
            "note": "skch::Map overlaps the host stage of batch i with the device stage of batch i+1 (pipelined); serial = no overlap"}


def csrc_sha16():
    """hash of the kernel sources (mashmap_amd/csrc/*.hip, *.h) as they lie: lets a line say whether committed PMC counters belong to this tree"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mashmap_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mashmap_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_entry(workload_key, kernel="k_sketch_fast"):
    """counters of one kernel from the committed PMC passes of this workload (profiles/pmc_traffic.json: per workload, per kernel,
    per launch), and where they came from -- they are NOT measured in the bench run itself; `pmc_age` says which tree they were taken on
    and whether the kernel sources have changed since"""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        wl = json.load(open(pmc)).get(workload_key)
        if not wl or kernel not in wl:
            return None, None
        then, now = wl.get("_csrc_sha16"), csrc_sha16()
        age = {"file": "profiles/pmc_traffic.json[%s]" % workload_key, "passes": "profiles/%s_pmc_*.csv" % wl.get("_source", "?"), "collected": wl.get("_collected", "?"),
               "csrc_sha16_then": then, "csrc_sha16_now": now, "kernel_sources_unchanged": (then == now) if then else None,
               "note": "rocprofv3 --pmc passes of this workload, committed; not measured in this run"}
        return wl[kernel], age
    except Exception:
        return None, None


def kernel_rooflines(ctx, W, workload_key, nF, prof, mean_points, pmc_ok):
    """one roofline object per kernel behind the sketch kernel (SURVEY section 8d's B_frag, term by term): algorithmic bytes per launch over the
    kernel's average HIP-event duration in the timed region, against the HBM peak; fabric traffic and VALU instructions per launch from the
    committed PMC passes of the same workload when there are any.  What actually binds each kernel is in `binds`."""
    SKETCH = W["sketch"]
    try:
        cnts = ctx.pass_counts()
    except Exception:
        cnts = {"l1": 0, "l2": 0, "queued": 0, "stream_entries": 0}
    ops = float(cnts["stream_entries"])
    spec = [
        ("k_lookup_l1", "lookup", (16.0 * SKETCH + 24.0 * mean_points) * nF,
         "16 B per sketch entry (hash in, table answer) + 24 B per interval point (SURVEY 8d) x fragments",
         "HBM bandwidth at 128-byte line granularity against a human-scale seed table (one line per probe); the Infinity Cache's gather rate against a 100 Mbp one (DESIGN.md section 3.3)"),
        ("k_l2_locate", "l2_locate", 20.0 * ops,
         "16 B read per index event a candidate touches + at most 4 B written per stream entry, x the %d entries reserved for the %d candidates' streams" % (int(ops), cnts["l1"]),
         "VALU issue + exposed latency (LDS sketch search per event); its slices of the index come mostly from the Infinity Cache (candidates are taken in reference order)"),
        ("k_l2_sweep", "l2", 4.0 * ops,
         "4 B per stream entry read by the lane that owns the candidate (state lives in LDS)",
         "VALU issue: ~110 straight-line lane-mask instructions per stream entry for 64 candidates at a time; LDS-limited occupancy at large sketches"),
    ]
    out = []
    for kern, key, bytes_launch, what, binds in spec:
        ms, n = prof.get(key, (0.0, 0))
        if not n or ms <= 0 or bytes_launch <= 0:
            continue
        avg = ms / n * (n / max(1, prof["sketch"][1]))          # several brackets per pass (e.g. extents + locate): per pass
        ach = bytes_launch / (avg / 1e3) / 1e9
        ent, age = pmc_entry(workload_key, kern) if pmc_ok else (None, None)
        out.append({"kernel": kern, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                    "avg_ms_per_pass": round(avg, 3), "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes": what, "binds": binds,
                    "traffic": ent.get("hbm_bytes_per_launch") if ent else None,
                    "valu_wave_instructions_per_launch": ent.get("SQ_INSTS_VALU") if ent else None, "pmc_age": age})
    return out


VALU_PEAK_GINST = SIMDS * CLOCK_HZ / 2.0 / 1e9   # wave64 VALU instructions per second the part can issue at the guide's nominal 2 cycles each (SIMD-32)
VALU_MIX_CYCLES = 3.35                           # cycles per wave-instruction of the sketch kernel's VOP3/VOP2 mix (profiles/r02_valu_rate.txt, r02_sketch_instruction_mix.txt)


def roofline_block(ctx, capi, W, workload_key, nF, prof, step_ms, pmc_ok, mean_points=0.0):
    """roofline of the dominant kernel (k_sketch_fast).  What binds it is VALU issue (2 x MurmurHash3_x64_128 per base, 0.25 B/bp in): `bound`
    is "valu", `achieved` = wave64 VALU instructions per launch (PMC SQ_INSTS_VALU of the committed passes of this workload) / the kernel's
    average HIP-event duration in THIS run, `peak` = 1024 SIMDs x 2.4 GHz / 2 cycles (MI355X_MICROARCH.md).  `hbm` is the secondary entry:
    algorithmic bytes per fragment = L/4 packed bases in + 24 B per sketch entry out over the same duration against 8 TB/s,
    and the PMC traffic.  `int` is the hash-only yardstick of SURVEY 8d(ii), measured in this run."""
    SEG, SKETCH = W["seg"], W["sketch"]
    sk_ms, sk_n = prof["sketch"]
    sk_avg = sk_ms / max(1, sk_n)
    frag_bytes = SEG / 4.0 + 24.0 * SKETCH
    ach = frag_bytes * nF / (sk_avg / 1e3) / 1e9 if sk_ms > 0 else 0.0
    # integer roofline: the same fragments through a kernel that only hashes (2 x MurmurHash3_x64_128 per base)
    integer = None
    try:
        hms = ctx.bench_hash_only(3)
        integer = {"hash_only_ms": round(hms, 3), "hash_only_gbps": round(nF * SEG / hms / 1e6, 2),
                   "sketch_kernel_frac": round(hms / sk_avg, 4) if sk_avg > 0 else None, "step_frac": round(hms / step_ms, 4),
                   "note": "k_hash_only: the sketch kernel's own geometry (positions per thread, threads per workgroup, LDS claim), staging and tables, "
                           "nothing but the two hashes per position; frac = its time / the kernel's (step's) time = share of the integer floor reached"}
    except capi.MashmapError as e:
        log("[bench] hash-only microbenchmark unavailable:", e)
    # HBM bytes / VALU instructions per launch of that kernel from the committed PMC passes of the same workload
    traffic = ninst = age = None
    if pmc_ok:
        ent, age = pmc_entry(workload_key)
        if ent:
            traffic = ent.get("hbm_bytes_per_launch")
            ninst = ent.get("SQ_INSTS_VALU")
    hbm = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
           "algorithmic_bytes_per_fragment": frag_bytes, "algorithmic_bytes_per_launch": frag_bytes * nF,
           "note": "small by construction: the kernel reads 0.25 B/bp and evaluates two full MurmurHash3_x64_128 per base"}
    kernels = kernel_rooflines(ctx, W, workload_key, nF, prof, mean_points, pmc_ok)
    out = {"kernel": "k_sketch_fast", "avg_launch_ms": round(sk_avg, 3), "traffic": traffic, "pmc_age": age, "hbm": hbm, "int": integer, "kernels": kernels}
    if ninst and sk_avg > 0:
        per_s = ninst / (sk_avg * 1e-3) / 1e9
        out.update({"bound": "valu", "achieved": round(per_s, 2), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave64 VALU instructions/s",
                    "frac": round(per_s / VALU_PEAK_GINST, 4),
                    "frac_vs_measured_mix": round(per_s / (VALU_PEAK_GINST * 2.0 / VALU_MIX_CYCLES), 4),
                    "valu_wave_instructions_per_launch": ninst,
                    "note": "VALU-issue bound kernel.  peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md nominal, what v_fma_f32 reaches); "
                            "frac_vs_measured_mix = against %.2f cycles per instruction, this kernel's mix of VOP3 integer ops (~4.2 cycles) and VOP2 ops (~2.3) as "
                            "scripts/probes/valu_rate.hip measures them on this part (profiles/r02_valu_rate.txt).  Instructions per launch from committed PMC passes "
                            "(pmc_age), duration from this run's HIP events" % VALU_MIX_CYCLES})
    else:                                    # no counters for this workload (scaled run): the HBM figures are all there is
        out.update({"bound": "hbm", "achieved": hbm["achieved"], "peak": hbm["peak"], "unit": hbm["unit"], "frac": hbm["frac"],
                    "note": "no committed SQ_INSTS_VALU pass for this workload: the HBM figures stand in; the kernel is VALU-issue bound (see `int`)"})
    return out


def load_batches(torch, dev, ctx, contigs, W, nreads, nb, seed, seq_base=0, keep_first=0):
    """nb distinct batches of the workload's reads (same shape, different reads) uploaded and packed on the device; the last one stays
    resident, the others are parked in slots 0 .. nb-2 (mm_reads_exchange).  Returns (fragments of the resident batch, the first
    `keep_first` reads of batch 0 as a host array or None)."""
    L = W["read_len"]
    offs = np.arange(nreads + 1, dtype=np.int64) * L
    first = None
    nF = 0
    for b in range(nb):
        if W.get("assembly"):                                  # the query is the reference's contigs, diverged and rearranged (configs[2])
            reads_t = torch.cat(make_assembly(torch, dev, contigs[:nreads], W["err"][0], seed=seed + 7919 * b))
            offs = np.cumsum([0] + [len(c) for c in contigs[:nreads]]).astype(np.int64)
        else:
            reads_t = make_reads(torch, dev, contigs, nreads, L, W["err"], seed=seed + 7919 * b)
        torch.cuda.synchronize()
        if b == 0 and keep_first:
            first = reads_t[:keep_first * L].cpu().numpy()
        nF = ctx.reads_upload_device(reads_t.data_ptr(), reads_t.numel(), offs, seqCounterBase=seq_base)
        del reads_t
        torch.cuda.empty_cache()
        if b < nb - 1:
            ctx.reads_exchange(b)
    return nF, first


class Rotation:
    """the resident batches taking turns: before pass i the resident batch goes to slot i % (nb - 1) and the one parked there becomes
    resident -- with nb batches and nb - 1 slots that visits all nb in a fixed cycle (nb = 3: C A B C A B ...)"""

    def __init__(self, ctx, nb):
        self.ctx, self.nb, self.i = ctx, nb, 0

    def next(self):
        if self.nb > 1:
            self.ctx.reads_exchange(self.i % (self.nb - 1))
        self.i += 1


def timed_passes(ctx, warmup, steps, nb=1):
    """W untimed passes, then K timed ones bracketed by a stream synchronisation, over nb resident batches in rotation; returns
    (seconds, per-kernel HIP-event times, {"steady": passes of the timed region that went through with one host wait, "redone": ...})"""
    rot = Rotation(ctx, nb)
    for _ in range(warmup):
        rot.next(); ctx.map()
    ctx.profile(True); ctx.profile_read(reset=True)
    ctx.synchronize()
    t0p = ctx.pass_totals()
    t0 = time.perf_counter()
    for _ in range(steps):
        rot.next(); ctx.map()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    t1p = ctx.pass_totals()
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    return dt, prof, {"timed": t1p["passes"] - t0p["passes"], "steady": t1p["steady"] - t0p["steady"], "redone": t1p["redone"] - t0p["redone"], "resident_batches": nb}


def human_scale_cpu_baseline(W, ref_np, reads_t, n_sample):
    """the stock binary (oracle/_ref/mashmap_ref, built from the reference's sources) on a sample of the target's reads against the SAME
    3 Gbp reference, defaults (it derives sketchSize 310 itself), on this box's host cores: index build and 'time spent mapping the query'"""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mashmap_ref")
    if not os.path.exists(ref_bin):
        return {"error": "oracle/_ref/mashmap_ref not here"}
    L = W["read_len"]
    sample = reads_t[:n_sample * L].cpu().numpy().reshape(n_sample, L)
    nt = max(4, min(64, 2 * usable_cpus()))
    with tempfile.TemporaryDirectory() as td:
        rp, qp, op = os.path.join(td, "ref.fa"), os.path.join(td, "q.fa"), os.path.join(td, "o.paf")
        write_fasta(rp, ["chr%d" % i for i in range(len(ref_np))], ref_np)
        write_fasta(qp, ["read%d" % i for i in range(n_sample)], list(sample), width=L)
        t0 = time.time()
        p = subprocess.run([ref_bin, "-r", rp, "-q", qp, "-o", op, "-t", str(nt)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.time() - t0
        tm = {}
        for line in p.stderr.splitlines():
            for key in ("computing the reference index", "mapping the query"):
                if "time spent " + key in line:
                    tm[key] = float(line.split(":")[-1].split()[0])
        lines = sum(1 for _ in open(op)) if os.path.exists(op) else 0
    if p.returncode != 0 or "mapping the query" not in tm:
        return {"error": "mashmap_ref exited with %d: %s" % (p.returncode, p.stderr[-300:])}
    log("[north_star] stock binary: index %.1f s, mapping %.2f s (%d reads, -t %d)" % (tm.get("computing the reference index", 0), tm["mapping the query"], n_sample, nt))
    return {"value": round(n_sample * L / tm["mapping the query"] / 1e9, 4), "unit": "Gbp/s", "cores": usable_cpus(), "threads": nt, "kind": "reference",
            "index_build_s": round(tm.get("computing the reference index", 0.0), 1), "wall_s": round(wall, 1), "paf_lines": lines,
            "sample": "%d of the target's reads (%.0f Mbp) vs the same %.0f Mbp reference written as FASTA; mashmap_ref (the reference's sources, GSL stand-in) with its "
                      "defaults, -t %d: %d host hardware threads, of which this process may use %d CPUs at once (container quota); 'time spent mapping the query' includes its "
                      "single-threaded FASTA reader" % (n_sample, n_sample * L / 1e6, sum(len(a) for a in ref_np) / 1e6, nt, os.cpu_count() or 1, usable_cpus())}


def north_star_target(torch, dev, capi, local, warmup, steps, cpu_reads=0, nb=3, repeat_rich=True, configs2=True):
    """BASELINE.json's north_star target sentence on one GPU: 1 M x 10 kbp ONT-like reads, pi 85, against a human-scale index (3 Gbp,
    24 x 125 Mbp), device-resident packed bases in -> candidate mappings out.  Two variants on the same data: the stock command line
    (segLength 5000, the metric's "s=5000": two fragments per read) and segLength 10000 ("10 kbp segments": one fragment per read);
    sketchSize 310 = what the stock binary derives for a 3 GB reference file at either segment length.  Then `repeat_rich`: the stock
    variant once more against a reference of the same size with human-like repeat structure (make_repeat_rich_reference)."""
    base = dict(WORKLOADS["northstar"])
    READ_LEN = base["read_len"]

    def measure(contigs, ref_np, W, wl_key, pmc_ok):
        nreads, READ_LEN = W["reads"], W["read_len"]
        ctx = capi.Context(k=W["k"], segLength=W["seg"], sketchSize=W["sketch"], flags=capi.MM_FLAG_HG_FILTER, device=local)
        t0 = time.time()
        ctx.index_build(ref_np, kmerPct=0.001)
        index_s = time.time() - t0
        ctx.set_tables_default(W["pi"])
        nF, _ = load_batches(torch, dev, ctx, contigs, W, nreads, nb, seed=1000)
        dt, prof, passes = timed_passes(ctx, warmup, steps, nb)
        step_ms = dt / steps * 1e3
        n1, n2 = ctx.result_counts()
        stats, _, _ = ctx.results()
        nmap = len(ctx.mappings())
        cnts = ctx.pass_counts()
        lay = ctx.index_layout() if hasattr(ctx, "index_layout") else None
        r = {"value": round(nreads * READ_LEN * steps / dt / 1e9, 4), "unit": "Gbp/s", "ms_per_step": round(step_ms, 3), "steps": steps, "warmup": warmup, "passes": passes,
             "kernels": {k: {"ms_per_step": v[0] / steps, "launches_per_step": v[1] / steps} for k, v in prof.items() if v[1]},
             "roofline": roofline_block(ctx, capi, W, wl_key, nF, prof, step_ms, pmc_ok, float(stats["nPoints"].mean())),
             "index_build_s": round(index_s, 2),
             "fragments": nF, "interval_points_per_fragment": round(float(stats["nPoints"].mean()), 1), "l1_candidates_per_fragment": round(n1 / max(1, nF), 3),
             "l2_loci_per_fragment": round(n2 / max(1, nF), 3), "candidate_mappings_per_fragment": round(nmap / max(1, nF), 3),
             "hard_list_share": round(cnts.get("hard", 0) / max(1, nF), 5), "hbm_point_path_share": round(cnts["queued"] / max(1, nF), 5),
             "workload": ("%d x %d bp " + ("assembly contigs (the reference + 1%% substitutions + rearrangements)" if W.get("assembly") else "reads (10%% ONT-like error)") +
                          " vs %.0f Mbp synthetic reference (%d contigs), k %d, segLength %d, sketchSize %d, pi %.2f: "
                          "%d fragments, %.1f interval points per fragment, %d L1 candidates, %d L2 loci, %d candidate mappings (last pass)")
                         % (nreads, READ_LEN, sum(len(a) for a in ref_np) / 1e6, len(ref_np), W["k"], W["seg"], W["sketch"], W["pi"], nF,
                            float(stats["nPoints"].mean()), n1, n2, nmap)}
        if lay:
            r["index_layout"] = lay
        ctx.close()
        return r

    t0 = time.time()
    contigs = make_reference(torch, dev, base["ref_contigs"], base["ref_contig_len"])
    ref_np = contiguous_views(torch, contigs)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    res = {}
    for key, seg in (("segLength5000", 5000), ("segLength10000", 10000)):
        res[key] = measure(contigs, ref_np, dict(base, seg=seg), "northstar" if seg == 5000 else "northstar_seg10000", True)
        log("[north_star] %s: %.1f Gbp/s, %.1f ms per pass, index %.1f s" % (key, res[key]["value"], res[key]["ms_per_step"], res[key]["index_build_s"]))
    out = dict(res["segLength5000"])
    if cpu_reads > 0:
        try:
            sample = make_reads(torch, dev, contigs, cpu_reads, READ_LEN, base["err"], seed=1000)
            out["cpu_baseline"] = human_scale_cpu_baseline(base, ref_np, sample, cpu_reads)
            del sample
        except Exception as e:
            log("[north_star] cpu_baseline failed:", repr(e)); out["cpu_baseline"] = {"error": repr(e)}
    out["what"] = ("BASELINE.json north_star target (>= 50 query Gbp/s sketch+map on 1 x MI355X, 10 kbp reads at pi 85 against a human-scale index), measured in "
                   "this run behind the headline configuration; inputs resident in HBM, same timed region as `value`")
    out["target_gbps"] = 50.0
    out["sketchSize_note"] = base["sketch_note"]
    out["synthetic_data_s"] = round(gen_s, 2)
    out["segLength_10000"] = res["segLength10000"]
    if configs2:
        # BASELINE configs[2] on the same reference: the 3 Gbp assembly against it, --pi 95 -s 10000 -J 40 -f one-to-one -- resident passes, then
        # FASTA -> PAF through the command line (`bench.py --workload configs2` is the same measurement by itself)
        try:
            W2 = dict(WORKLOADS["configs2"])
            c2 = measure(contigs, ref_np, W2, "configs2", True)
            c2["e2e"] = e2e_assembly(torch, dev, W2, contigs, max(4, min(128, os.cpu_count() or 1)))
            out["configs2"] = c2
            log("[north_star] configs2: %.1f Gbp/s resident, %.1f ms per pass; FASTA -> PAF %s" % (c2["value"], c2["ms_per_step"], {k: c2["e2e"].get(k) for k in ("value", "map_s", "error")}))
        except Exception as e:
            log("[north_star] configs2 failed:", repr(e)); out["configs2"] = {"error": repr(e)}
    del contigs, ref_np
    torch.cuda.empty_cache()
    if repeat_rich:
        try:
            t0 = time.time()
            contigs, summary = make_repeat_rich_reference(torch, dev, base["ref_contigs"], base["ref_contig_len"])
            ref_np = contiguous_views(torch, contigs)
            torch.cuda.synchronize()
            rr = measure(contigs, ref_np, dict(base), "northstar_repeat_rich", True)
            rr["reference"] = summary
            rr["synthetic_data_s"] = round(time.time() - t0, 2)
            rr["what"] = ("the north_star target workload (same reads-per-batch, read length, error model, parameters) with reference AND reads drawn from a 3 Gbp sequence "
                          "with human-like repeat structure instead of i.i.d. uniform ACGT; PAF parity at this shape: tests/test_gpu_zz_humanscale.py::test_repeat_rich_reference")
            out["repeat_rich"] = rr
            log("[north_star] repeat_rich: %.1f Gbp/s, %.1f ms per pass, %.2f L1 candidates per fragment, hard-list share %.4f" %
                (rr["value"], rr["ms_per_step"], rr["l1_candidates_per_fragment"], rr["hard_list_share"]))
        except Exception as e:
            log("[north_star] repeat_rich failed:", repr(e)); out["repeat_rich"] = {"error": repr(e)}
    return out


class StubContext:
    """CPU stand-in used ONLY by the CPU tests of this script's rank loop (--stub: tests/test_bench_spawn.py): the same StepLoop drives it
    as drives a capi.Context, over gloo.  It maps nothing; a line produced with it says "data": "stub".  The exchange has the library's
    shape -- _begin snapshots this rank's records, _end runs the collective (an all-gatherv through mashmap_amd/shard.py's plan) -- and
    every call is logged with a timestamp (MM_STUB_LOG) so that a test can check the order the loop issues them in.
    MM_STUB_EMPTY_RANK=r: rank r has no mappings at all; MM_STUB_FAIL_END=r:k: rank r's k-th _end raises."""

    def __init__(self, rank, dist=None):
        self.rank, self.dist = rank, dist
        self.pending = None
        self.ends = 0
        self.npass = 0
        self.logf = open(os.environ["MM_STUB_LOG"] + ".%d" % rank, "w") if os.environ.get("MM_STUB_LOG") else None
        self.empty = os.environ.get("MM_STUB_EMPTY_RANK") == str(rank)
        fe = os.environ.get("MM_STUB_FAIL_END", "")
        self.fail_end = int(fe.split(":")[1]) if fe and fe.split(":")[0] == str(rank) else None

    def _log(self, what, **kw):
        if self.logf:
            self.logf.write(json.dumps(dict(t=time.perf_counter(), ev=what, **kw)) + "\n"); self.logf.flush()

    def reads_exchange(self, slot):
        self._log("exchange", slot=slot)

    def map(self):
        time.sleep(0.002)
        self.npass += 1
        self._log("map", n=self.npass)

    def _records(self):
        from mashmap_amd import capi
        mine = np.zeros(0 if self.empty else 3 + self.rank, dtype=capi.MAPPING_DT)
        mine["querySeqId"] = self.rank
        mine["fragStart"] = self.npass
        return mine

    def _gather(self, mine):
        from mashmap_amd import shard
        got, counts = shard.allgatherv_mappings(mine, self.dist)
        assert len(got) == sum(counts) and (got["querySeqId"] == np.repeat(np.arange(len(counts)), counts)).all()
        return got, counts

    def allgatherv_mappings(self):
        self._log("sync_gather"); self._gather(self._records())

    def allgatherv_mappings_begin(self):
        assert self.pending is None, "two exchanges in flight"
        self.pending = self._records()
        self._log("begin", n=self.npass)

    def allgatherv_mappings_end(self):
        assert self.pending is not None, "_end without _begin"
        self.ends += 1
        if self.fail_end is not None and self.ends == self.fail_end:
            raise RuntimeError("stub: rank %d fails in its exchange %d" % (self.rank, self.ends))
        got, counts = self._gather(self.pending)
        of = sorted(set(int(x) for x in got["fragStart"]))          # every rank snapshots after the same pass: one pass number in the gathered records
        assert len(of) <= 1, "records of several passes in one exchange: %s" % of
        self._log("end", of=of[0] if of else -1, counts=[int(c) for c in counts])
        self.pending = None

    def synchronize(self):
        pass

    def profile(self, on):
        pass

    def profile_read(self, reset=False):
        return {}

    def pass_totals(self):
        return {"passes": self.npass, "steady": max(0, self.npass - 1), "redone": 0}

    def close(self):
        if self.logf:
            self.logf.close()


class StepLoop:
    """the timed loop of a rank: W warm-up steps, then exactly K timed ones between two fences.  A step = next resident batch in
    (Rotation) + one pass of the hot path + -- N > 1 -- the exchange of that pass's candidate mappings: by default overlapped, i.e. _end of
    the previous step's exchange and _begin of this one's (it then runs on the library's exchange stream under the kernels of the next
    pass); --sync-exchange: the blocking form.  A fence waits for the exchange still in flight, meets the other ranks (barrier) and
    drains the device, so the last exchange is INSIDE the timed region; the elapsed time is the max over ranks."""

    def __init__(self, ctx, world, dist, torch, nb, sync_exchange=False, log_event=None):
        self.ctx, self.world, self.dist, self.torch, self.sync_exchange = ctx, world, dist, torch, sync_exchange
        self.rot = Rotation(ctx, nb)
        self.inflight = False
        self.log_event = log_event or (lambda *_: None)

    def step(self):
        self.rot.next()
        self.ctx.map()
        if self.world > 1:
            if self.sync_exchange:
                self.ctx.allgatherv_mappings()
                return
            if self.inflight:
                self.ctx.allgatherv_mappings_end()
            self.ctx.allgatherv_mappings_begin()
            self.inflight = True

    def fence(self):
        if self.inflight:
            self.ctx.allgatherv_mappings_end()
            self.inflight = False
        if self.world > 1:
            self.dist.barrier()
        if self.torch.cuda.is_available():
            self.torch.cuda.synchronize()
        self.ctx.synchronize()

    def run(self, warmup, steps, device=None):
        """returns (seconds = max over ranks, per-kernel HIP-event times, passes dict of the timed region)"""
        for _ in range(warmup):
            self.step()
        self.ctx.profile(True); self.ctx.profile_read(reset=True)
        self.fence()
        p0 = self.ctx.pass_totals()
        self.log_event("t0")
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        dt = time.perf_counter() - t0
        self.log_event("t1")
        p1 = self.ctx.pass_totals()
        prof = self.ctx.profile_read(reset=True)
        self.ctx.profile(False)
        if self.world > 1:
            tmax = self.torch.tensor([dt], dtype=self.torch.float64, device=device if device is not None else "cpu")
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            dt = float(tmax.item())
        passes = {"timed": p1["passes"] - p0["passes"], "steady": p1["steady"] - p0["steady"], "redone": p1["redone"] - p0["redone"], "resident_batches": self.rot.nb}
        return dt, prof, passes


LINE_LIMIT = 6000   # the driver keeps the last 8 000 characters of stdout: the final line has to fit with room to spare


def _r(x, n=4):
    return round(x, n) if isinstance(x, float) else x[:160] if isinstance(x, str) else x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _compact_roofline(r):
    """numbers only: what the driver and the review read; the prose stays in the full record"""
    if not isinstance(r, dict):
        return None
    out = _pick(r, ("kernel", "avg_launch_ms", "bound", "achieved", "peak", "unit", "frac", "frac_vs_measured_mix", "traffic", "valu_wave_instructions_per_launch"))
    hbm = r.get("hbm") or {}
    out["hbm"] = _pick(hbm, ("achieved", "peak", "unit", "frac", "traffic"))
    out["algorithmic_bytes_per_launch"] = hbm.get("algorithmic_bytes_per_launch")
    out["int"] = _pick(r.get("int") or {}, ("hash_only_ms", "sketch_kernel_frac", "step_frac"))
    age = r.get("pmc_age") or {}
    out["pmc_tree"] = {"passes": age.get("passes"), "csrc_sha16": age.get("csrc_sha16_then"), "unchanged": age.get("kernel_sources_unchanged")} if age else None
    out["kernels"] = [_pick(k, ("kernel", "avg_ms_per_pass", "achieved", "frac", "traffic", "algorithmic_bytes_per_launch")) for k in r.get("kernels") or []]
    return out


def _compact_side(d):
    """one side measurement (north_star variants, configs2): value, time, the shares the review asks for, and its dominant kernel's fractions"""
    if not isinstance(d, dict):
        return None
    if "error" in d and "value" not in d:
        return {"error": str(d["error"])[:160]}
    out = _pick(d, ("value", "ms_per_step", "hbm_point_path_share", "hard_list_share", "l1_candidates_per_fragment", "index_build_s"))
    p = d.get("passes") or {}
    if p:
        out["redone"] = p.get("redone")
    r = d.get("roofline") or {}
    if r:
        out["roofline"] = {"frac": r.get("frac"), "hbm_frac": (r.get("hbm") or {}).get("frac"), "sketch_kernel_frac": (r.get("int") or {}).get("sketch_kernel_frac"),
                           "pmc_unchanged": (r.get("pmc_age") or {}).get("kernel_sources_unchanged")}
    k = d.get("kernels") or {}
    if k:
        out["kernels_ms"] = {n: round(v["ms_per_step"], 2) for n, v in k.items() if v["ms_per_step"] >= 0.05}
    return out


def compact_line(full, full_path=None):
    """the ONE line the driver parses, from the full record of a run: every key of the bench contract, `roofline` and `cpu_baseline`
    as numbers, the side measurements as one small object each -- no prose.  Everything else is in `full_path`."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c = full.get("config") or {}
    cfg = _pick(c, ("workload", "k", "segLength", "sketchSize", "percentageIdentity", "fragments_per_gpu", "resident_batches", "mean_interval_points_per_fragment",
                    "l1_candidates_per_gpu", "l2_loci_per_gpu", "candidate_mappings_per_gpu", "hard_list_fragments", "index_build_s", "host_synchronisations_last_pass"))
    cfg["workload"] = str(cfg.get("workload", ""))[:200]
    cfg["parallelism"] = "single GPU" if full.get("n_gpus", 1) == 1 else "reads sharded, index replicated, RCCL all-gatherv of candidate mappings"
    if c.get("rccl"):
        cfg["rccl"] = _pick(c["rccl"], ("world_seen", "error"))
    line["config"] = cfg
    line["passes"] = _pick(full.get("passes") or {}, ("timed", "steady", "redone", "resident_batches"))
    line["roofline"] = _compact_roofline(full.get("roofline"))
    line["kernels_ms"] = {n: round(v["ms_per_step"], 3) for n, v in (full.get("kernels") or {}).items()}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "threads", "kind"))
        if "sample" in cb:
            line["cpu_baseline"]["sample"] = str(cb["sample"])[:100]
        fc = cb.get("fragment_compute") or {}
        if fc:
            line["cpu_baseline"]["gbps_per_core"] = fc.get("gbps_per_core")
        if "error" in cb:
            line["cpu_baseline"]["error"] = str(cb["error"])[:160]
    hp = full.get("host_path")
    if isinstance(hp, dict):
        line["host_path"] = _pick(hp, ("device_ms", "download_ms", "host_ms", "host_threads", "gbps_pipelined", "error"))
    e = full.get("e2e")
    if isinstance(e, dict):
        line["e2e"] = _pick(e, ("value", "map_s", "index_s", "paf_lines", "error"))
        st = e.get("stages") or {}
        line["e2e"].update(_pick(st, ("reader_s", "device_stage_s", "post_s")))
    ns = full.get("north_star_target")
    if isinstance(ns, dict):
        t = _compact_side(ns) or {}
        t["target_gbps"] = ns.get("target_gbps", 50.0)
        t["seg10000"] = _compact_side(ns.get("segLength_10000"))
        t["repeat_rich"] = _compact_side(ns.get("repeat_rich"))
        cpu = ns.get("cpu_baseline")
        t["cpu"] = _pick(cpu, ("value", "cores", "threads", "kind", "index_build_s", "error")) if isinstance(cpu, dict) else None
        line["north_star_target"] = t
    c2 = full.get("configs2")
    if isinstance(c2, dict):
        line["configs2"] = _compact_side(c2)
        if isinstance(c2.get("e2e"), dict) and line["configs2"] is not None:
            line["configs2"]["e2e"] = _pick(c2["e2e"], ("value", "map_s", "paf_lines", "error"))
    if full_path:
        line["full"] = full_path
    # whatever a future key adds, the line is never allowed past the limit: shed the optional objects, largest first
    for drop in ("kernels_ms", "host_path", "e2e", "configs2"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) >= LINE_LIMIT:
        for t in (line.get("north_star_target") or {}, (line.get("north_star_target") or {}).get("seg10000") or {}, (line.get("north_star_target") or {}).get("repeat_rich") or {}):
            t.pop("kernels_ms", None)
        (line.get("roofline") or {}).pop("kernels", None)
    return line


def emit(full):
    """stdout of rank 0: the full record goes to profiles/bench_last_full.json (and gpurun_out/, which travels back from a GPU box), the side
    measurements one small object per line, and LAST the line the driver parses."""
    paths = []
    for d in ("profiles", "gpurun_out"):
        try:
            os.makedirs(os.path.join(ROOT, d), exist_ok=True)
            with open(os.path.join(ROOT, d, "bench_last_full.json"), "w") as f:
                json.dump(full, f, indent=1)
            paths.append("%s/bench_last_full.json" % d)
        except OSError as e:
            log("[bench] could not write the full record under %s/: %r" % (d, e))
    line = compact_line(full, paths[0] if paths else None)
    for key in ("host_path", "e2e", "configs2"):
        if key in line:
            print(json.dumps({"side": key, **line[key]}), flush=True)
    ns = line.get("north_star_target")
    if ns:
        print(json.dumps({"side": "north_star_target", **{k: v for k, v in ns.items() if k not in ("seg10000", "repeat_rich")}}), flush=True)
        for key in ("seg10000", "repeat_rich"):
            if ns.get(key):
                print(json.dumps({"side": "north_star_target." + key, **ns[key]}), flush=True)
    txt = json.dumps(line)
    assert len(txt) < LINE_LIMIT, "bench line is %d characters" % len(txt)
    print(txt, flush=True)


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def run_cli_staged(exe, argv, reps=2, log_env="MM_E2E_LOG"):
    """runs the mashmap_hip command line `reps` times with MASHMAP_HIP_TIMING=1 (the second run finds the files in the page cache) and returns
    the best run's 'time spent' figures and the rows of its stage log"""
    import re
    env = dict(os.environ, MASHMAP_HIP_TIMING="1")
    best = None
    for rep in range(reps):
        t0 = time.time()
        p = subprocess.run([exe] + argv, capture_output=True, text=True, env=env)
        wall = time.time() - t0
        if p.returncode != 0:
            return {"error": "mashmap_hip exited with %d: %s" % (p.returncode, p.stderr[-400:])}
        tmap = float(re.search(r"time spent mapping the query\s*:\s*([0-9.eE+-]+)", p.stderr).group(1))
        tidx = float(re.search(r"time spent computing the reference index\s*:\s*([0-9.eE+-]+)", p.stderr).group(1))
        dev_rows = re.findall(r"device stage \(.*?download of (\d+) candidate mappings\): ([0-9.eE+-]+) s \(upload ([0-9.eE+-]+), kernels ([0-9.eE+-]+), download ([0-9.eE+-]+)\)(?: \[bases (\d+)\])?", p.stderr)
        rd_rows = re.findall(r"reader: parsed (\d+) records, (\d+) bases in ([0-9.eE+-]+) s", p.stderr)
        post = [float(x) for x in re.findall(r"post stage: chain \+ filter \+ format ([0-9.eE+-]+) s", p.stderr)]
        outp = [float(x) for x in re.findall(r", output ([0-9.eE+-]+) s", p.stderr)]
        final = [float(x) for x in re.findall(r"one-to-one filter \+ output ([0-9.eE+-]+) s", p.stderr)]
        cur = dict(map_s=tmap, index_s=tidx, wall_s=wall, stderr=p.stderr, dev_rows=dev_rows, rd_rows=rd_rows, post_s=sum(post), output_s=sum(outp), final_s=sum(final))
        if best is None or tmap < best["map_s"]:
            best = cur
    if os.environ.get(log_env):                            # the stage log of the best run, for profiles/
        with open(os.environ[log_env], "w") as f:
            f.write("\n".join(l for l in best["stderr"].splitlines() if "timing" in l or "time spent" in l) + "\n")
    return best


def stage_summary(best):
    dev_rows = best["dev_rows"]
    return {"reader_s": round(sum(float(r[2]) for r in best["rd_rows"]), 4), "reader_batches": len(best["rd_rows"]),
            "device_stage_s": round(sum(float(r[1]) for r in dev_rows), 4), "device_upload_wait_s": round(sum(float(r[2]) for r in dev_rows), 4),
            "device_kernels_s": round(sum(float(r[3]) for r in dev_rows), 4), "device_download_s": round(sum(float(r[4]) for r in dev_rows), 4), "device_passes": len(dev_rows),
            "post_s": round(best["post_s"], 4), "output_s": round(best["output_s"], 4), "final_filter_s": round(best.get("final_s", 0.0), 4)}


def e2e_assembly(torch, dev, W, contigs, threads, stock=False, keep_dir=None):
    """configs[2] through the command line: the 3 Gbp reference and the assembly (make_assembly of the same contigs) written as FASTA,
    `mashmap_hip --pi 95 -s 10000 -f one-to-one -J 40`, its 'time spent mapping the query' and stage log; with `stock` the reference's
    own binary on the same files, PAF bytes compared."""
    import shutil
    exe = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
    if not os.path.exists(exe):
        return {"error": "mashmap_amd/lib/mashmap_hip not built"}
    td = keep_dir or tempfile.mkdtemp(prefix="mm_e2e2_")
    try:
        bases = sum(len(c) for c in contigs)
        if shutil.disk_usage(td).free < 2.1 * bases + (2 << 30):
            return {"error": "only %.1f GB free under %s" % (shutil.disk_usage(td).free / 1e9, td)}
        rp, qp, op = os.path.join(td, "ref.fa"), os.path.join(td, "asm.fa"), os.path.join(td, "out.paf")
        t0 = time.time()
        write_fasta(rp, ["chr%d" % i for i in range(len(contigs))], [c.cpu().numpy() for c in contigs])
        asm = make_assembly(torch, dev, contigs, W["err"][0], seed=2021)
        write_fasta(qp, ["ctg%d" % i for i in range(len(asm))], [c.cpu().numpy() for c in asm])
        del asm
        torch.cuda.empty_cache()
        write_s = time.time() - t0
        argv = ["-s", str(W["seg"]), "--pi", str(int(round(W["pi"] * 100))), "-J", str(W["sketch"])] + list(W.get("cli", []))
        best = run_cli_staged(exe, ["-r", rp, "-q", qp, "-o", op, "-t", str(threads)] + argv, log_env="MM_E2E2_LOG")
        if "error" in best:
            return best
        lines = sum(1 for _ in open(op, "rb"))
        out = {"what": "mashmap_hip -r ref.fa -q asm.fa %s (FASTA -> PAF): %d contigs, %.2f Gbp assembly vs the %.2f Gbp reference it was derived from" % (" ".join(argv), len(contigs), bases / 1e9, bases / 1e9),
               "value": round(bases / best["map_s"] / 1e9, 3), "unit": "Gbp/s", "map_s": round(best["map_s"], 4), "index_s": round(best["index_s"], 3), "wall_s": round(best["wall_s"], 3),
               "paf_lines": lines, "threads": threads, "usable_cpus": usable_cpus(), "fasta_write_s": round(write_s, 1), "stages": stage_summary(best)}
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "mashmap_ref")
        if stock and os.path.exists(ref_bin):
            sp = os.path.join(td, "stock.paf")
            nt = max(4, min(64, 2 * usable_cpus()))
            t0 = time.time()
            p = subprocess.run([ref_bin, "-r", rp, "-q", qp, "-o", sp, "-t", str(nt)] + argv, capture_output=True, text=True)
            tm = {}
            for line in p.stderr.splitlines():
                for key in ("computing the reference index", "mapping the query"):
                    if "time spent " + key in line:
                        tm[key] = float(line.split(":")[-1].split()[0])
            out["stock"] = {"rc": p.returncode, "threads": nt, "wall_s": round(time.time() - t0, 1), "index_s": tm.get("computing the reference index"), "map_s": tm.get("mapping the query"),
                            "value": round(bases / tm["mapping the query"] / 1e9, 4) if "mapping the query" in tm else None,
                            "paf_identical": p.returncode == 0 and open(sp, "rb").read() == open(op, "rb").read()}
        return out
    finally:
        if not keep_dir:
            shutil.rmtree(td, ignore_errors=True)


def e2e_fasta_to_paf(torch, dev, W, ref_np, nreads, threads):
    """the path a user runs, inside this run: the `mashmap_hip` command line (skch::Sketch + skch::Map on the C ABI) on the workload's
    FASTA files -- parse + pack, upload, kernels, download, chaining + filters, PAF text --, its own 'time spent mapping the query' and
    the per-stage seconds of its MASHMAP_HIP_TIMING log.  The FASTA is written first (reads regenerated with the headline's seed)."""
    import re
    import shutil
    exe = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
    if not os.path.exists(exe):
        return {"error": "mashmap_amd/lib/mashmap_hip not built"}
    L = W["read_len"]
    td = tempfile.mkdtemp(prefix="mm_e2e_")
    try:
        need = nreads * (L + 14) + sum(len(a) for a in ref_np) * 1.02 + (1 << 30)
        free = shutil.disk_usage(td).free
        scaled = None
        if free < need:
            scaled = max(1000, int(nreads * (free - (2 << 30)) / need))
            if free < (3 << 30):
                return {"error": "only %.1f GB free under %s" % (free / 1e9, td)}
            nreads = scaled
        rp, qp, op = os.path.join(td, "ref.fa"), os.path.join(td, "reads.fa"), os.path.join(td, "out.paf")
        t0 = time.time()
        write_fasta(rp, ["chr%d" % i for i in range(len(ref_np))], ref_np)
        contigs = [torch.from_numpy(a).to(dev) for a in ref_np]
        with open(qp, "wb") as f:
            chunk = 100_000
            for r0 in range(0, nreads, chunk):
                n = min(chunk, nreads - r0)
                rd = make_reads(torch, dev, contigs, n, L, W["err"], seed=5000 + r0).cpu().numpy().reshape(n, L)
                hdr = np.frombuffer(b"".join(b">read%07d\n" % (r0 + i) for i in range(n)), dtype=np.uint8).reshape(n, 13)      # fixed-width names
                f.write(np.concatenate([hdr, rd, np.full((n, 1), 10, dtype=np.uint8)], axis=1).tobytes())
        del contigs
        torch.cuda.empty_cache()
        write_s = time.time() - t0
        best = run_cli_staged(exe, ["-r", rp, "-q", qp, "-o", op, "-t", str(threads), "-s", str(W["seg"]), "--pi", str(int(round(W["pi"] * 100))), "-J", str(W["sketch"])])
        if "error" in best:
            return best
        bases = nreads * L
        dev_rows = best["dev_rows"]
        # a device-stage row covers one pass over one or several reader batches: its bases are in the row (skch_map.hpp), else the reader's batches in order
        pass_bases = [int(r[5]) for r in dev_rows if r[5]]
        if len(pass_bases) != len(dev_rows):
            pass_bases = [int(r[1]) for r in best["rd_rows"]][:len(dev_rows)]
        kern = [float(r[3]) for r in dev_rows]
        full = max(pass_bases) if pass_bases else 0
        fb = [(b, k) for b, k in zip(pass_bases, kern) if b >= 0.9 * full]
        lines = sum(1 for _ in open(op, "rb"))
        return {"what": "mashmap_hip -r ref.fa -q reads.fa -o out.paf (FASTA -> PAF) on this workload's files: %d x %d bp reads (%.2f GB of FASTA) vs %.0f Mbp; "
                        "'time spent mapping the query' = parse + pack + upload + kernels + download + chain/filter + PAF text, the three stages (reader | device | post) "
                        "overlapped on successive batches; best of two runs" % (nreads, L, os.path.getsize(qp) / 1e9, sum(len(a) for a in ref_np) / 1e6),
                "value": round(bases / best["map_s"] / 1e9, 3), "unit": "Gbp/s", "map_s": round(best["map_s"], 4), "index_s": round(best["index_s"], 3), "wall_s": round(best["wall_s"], 3),
                "paf_lines": lines, "threads": threads, "usable_cpus": usable_cpus(), "scaled_to_reads": scaled, "fasta_write_s": round(write_s, 1),
                "stages": stage_summary(best),
                "device_stage": {"gbps_kernels_all_passes": round(sum(pass_bases) / max(1e-9, sum(kern)) / 1e9, 2),
                                 "gbps_kernels_full_size_passes": round(sum(b for b, _ in fb) / max(1e-9, sum(k for _, k in fb)) / 1e9, 2) if fb else None,
                                 "full_size_passes": len(fb), "bases_per_pass": pass_bases,
                                 "note": "kernels = mm_map_fragments of a pass (sketch .. selection, its host waits included); a pass covers as many parsed batches as were "
                                         "waiting, up to MASHMAP_HIP_COALESCE_MBP per GPU (skch_map.hpp); full-size passes = those within 10 % of the largest"}}
    finally:
        shutil.rmtree(td, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="configs1")
    ap.add_argument("--batches", type=int, default=3, help="distinct resident batches taking turns in the timed loop (1..5)")
    ap.add_argument("--reads", type=int, default=int(os.environ.get("MM_BENCH_READS", 0)), help="reads per GPU and batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="default run only: skip the FASTA -> PAF run of the mashmap_hip command line")
    ap.add_argument("--no-north-star", action="store_true", help="default run only: skip the north_star target measurements (3 Gbp index) behind the headline one")
    ap.add_argument("--no-repeat-rich", action="store_true", help="default run only: skip north_star_target.repeat_rich")
    ap.add_argument("--no-configs2", action="store_true", help="default run only: skip the configs[2] measurement (3 Gbp assembly vs the 3 Gbp reference) behind the north_star ones")
    ap.add_argument("--north-star-steps", type=int, default=6, help="timed passes of each north_star variant (at most --steps)")
    ap.add_argument("--ref-contigs", type=int, default=0, help="contigs of the synthetic reference (default: the workload's)")
    ap.add_argument("--ref-contig-len", type=int, default=0)
    ap.add_argument("--kmer", type=int, default=0, help="k-mer size (default: the reference's 19; other sizes are not the BASELINE configuration)")
    ap.add_argument("--seg", type=int, default=0, help="segLength (default: the workload's; `--workload northstar --seg 10000` is the north_star sentence's 10 kbp segments, "
                                                       "sketchSize unchanged, as north_star_target.segLength_10000 measures it)")
    ap.add_argument("--cpu-sample", type=int, default=30000)
    ap.add_argument("--sync-exchange", action="store_true", help="N>1: all-gatherv on the compute stream instead of overlapped with the next batch")
    ap.add_argument("--repeat-rich-reference", action="store_true", help="draw reference and reads from make_repeat_rich_reference instead of uniform ACGT (not a BASELINE configuration)")
    ap.add_argument("--stock", action="store_true", help="configs2: also run the stock binary on the same FASTA files (index ~30 s + mapping) and compare the PAF bytes")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--north-star-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if not 1 <= args.batches <= 5:
        raise SystemExit("--batches must be 1..5 (MM_BATCH_SLOTS + 1)")

    if args.north_star_child:                           # the default run's second measurement (see north_star_target), one GPU
        import torch
        from mashmap_amd import capi
        torch.cuda.set_device(0)
        print(json.dumps(north_star_target(torch, torch.device("cuda", 0), capi, 0, args.warmup, args.steps, 0 if args.no_cpu_baseline else args.cpu_sample,
                                           nb=args.batches, repeat_rich=not args.no_repeat_rich, configs2=not args.no_configs2)), flush=True)
        return

    # ---- N ranks: start them ourselves unless a launcher already did
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        log("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd[2:9])))
        raise SystemExit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to report a number for a different GPU count" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    W = dict(WORKLOADS[args.workload])
    scaled = []
    if args.reads: W["reads"] = args.reads; scaled.append("reads")
    if args.ref_contigs: W["ref_contigs"] = args.ref_contigs; scaled.append("ref-contigs")
    if args.ref_contig_len: W["ref_contig_len"] = args.ref_contig_len; scaled.append("ref-contig-len")
    if args.kmer: W["k"] = args.kmer; scaled.append("kmer")
    if args.repeat_rich_reference: scaled.append("repeat-rich reference")
    wl_key = args.workload
    if args.seg and args.seg != W["seg"]:
        W["seg"] = args.seg; W["label"] += ", segLength %d" % args.seg
        wl_key = "%s_seg%d" % (args.workload, args.seg)
    if W.get("assembly"):                               # the query is the reference's contigs, diverged and rearranged: same count, same lengths
        W["reads"], W["read_len"] = min(W["reads"], W["ref_contigs"]), W["ref_contig_len"]
    is_default = args.workload == "configs1" and not scaled and wl_key == args.workload
    nb = args.batches

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.stub else "nccl", rank=rank, world_size=world)

    if args.stub:                                       # the real rank loop over a context that maps nothing (CPU tests)
        ctx = StubContext(rank, dist)
        loop = StepLoop(ctx, world, dist, torch, nb, args.sync_exchange, log_event=lambda ev: ctx._log(ev))
        dt, _, passes = loop.run(args.warmup, args.steps)
        if rank == 0:
            print(json.dumps({"metric": "query Gbp/s sketch+L1/L2 map (pi=85, s=5000)", "value": 0.0, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "u64", "data": "stub", "passes": passes, "config": {"workload": "STUB: no kernels ran (launcher test)"}}), flush=True)
        ctx.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from mashmap_amd import capi
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    K, SEG, SKETCH, PI, READ_LEN = W["k"], W["seg"], W["sketch"], W["pi"], W["read_len"]
    nreads = W["reads"]

    t0 = time.time()
    ref_summary = None
    if args.repeat_rich_reference:
        contigs, ref_summary = make_repeat_rich_reference(torch, dev, W["ref_contigs"], W["ref_contig_len"])
    else:
        contigs = make_reference(torch, dev, W["ref_contigs"], W["ref_contig_len"])
    ref_np = contiguous_views(torch, contigs)
    torch.cuda.synchronize()
    log("[rank %d] synthetic reference: %.1f s" % (rank, time.time() - t0))

    ctx = capi.Context(k=K, segLength=SEG, sketchSize=SKETCH, flags=capi.MM_FLAG_HG_FILTER, device=local)
    t0 = time.time()
    ctx.index_build(ref_np, kmerPct=0.001)
    index_s = time.time() - t0
    t0 = time.time()
    ctx.set_tables_default(PI)
    log("[rank %d] index build: %.1f s; integer tables: %.2f s" % (rank, index_s, time.time() - t0))
    # the CPU leg indexes the reference with the stock binary: minutes beyond a few hundred Mbp, so it rides on the default workload only
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and sum(len(a) for a in ref_np) <= 400e6
    want_e2e = is_default and rank == 0 and world == 1 and not args.no_e2e
    want_e2e2 = bool(W.get("assembly")) and rank == 0 and world == 1 and not args.no_e2e
    e2e2 = None
    if want_e2e2:                                       # before the reads take the room: needs the reference on the device once more
        try:
            e2e2 = e2e_assembly(torch, dev, W, contigs, max(4, min(128, os.cpu_count() or 1)), stock=args.stock)
            log("[bench] configs[2] FASTA -> PAF: %s" % {k: e2e2.get(k) for k in ("value", "map_s", "stages", "stock", "error")})
        except Exception as e:
            log("[bench] e2e_assembly failed:", repr(e)); e2e2 = {"error": repr(e)}
    t0 = time.time()
    nF, reads_np = load_batches(torch, dev, ctx, contigs, W, nreads, nb, seed=1000 + rank, seq_base=rank * nreads,
                                keep_first=min(nreads, args.cpu_sample) if want_cpu else 0)
    log("[rank %d] %d batches of %d reads generated, packed and parked: %.1f s" % (rank, nb, nreads, time.time() - t0))
    del contigs
    ref_lens = [len(a) for a in ref_np]
    if not (want_cpu or want_e2e):
        ref_np = None
    torch.cuda.empty_cache()

    rccl = None
    if world > 1:                                       # the product's RCCL communicator: the id travels through torch's store
        box = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init_rank(box[0], rank, world)
        try:
            rccl = ctx.comm_info()                      # what the communicator itself reports: ranks seen (ncclCommCount), library bound
        except Exception as e:
            rccl = {"error": repr(e)}

    loop = StepLoop(ctx, world, dist, torch, nb, args.sync_exchange)
    dt, prof, passes = loop.run(args.warmup, args.steps, device=dev)
    n1, n2 = ctx.result_counts()

    if rank == 0:
        stats, _, _ = ctx.results()
        nmap = len(ctx.mappings())
        cnts = ctx.pass_counts()
        bases_step = nreads * READ_LEN * world
        value = bases_step * args.steps / dt / 1e9
        step_ms = dt / args.steps * 1e3
        P = float(stats["nPoints"].mean())
        roofline = roofline_block(ctx, capi, W, wl_key, nF, prof, step_ms, not scaled, P)
        kernels = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps} for k, v in prof.items() if v[1]}
        ref_mbp = sum(ref_lens) / 1e6
        out = {
            "metric": "query Gbp/s sketch+L1/L2 map (pi=85, s=5000)", "value": round(value, 4), "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("%s%s: %d x %d bp assembly contigs/GPU (the reference + %s substitutions + 1-5 Mbp inversions / translocations) vs %.0f Mbp synthetic reference (%d contigs)"
                                    if W.get("assembly") else "%s%s: %d x %d bp reads/GPU (%s ONT-like error) vs %.0f Mbp synthetic reference (%d contigs)")
                                   % (W["label"], " SCALED (%s)" % ", ".join(scaled) if scaled else "", nreads, READ_LEN,
                                      "%.0f%%" % (W["err"][0] * 100) if W["err"][0] == W["err"][1] else "%.0f-%.0f%%" % (W["err"][0] * 100, W["err"][1] * 100),
                                      ref_mbp, len(ref_lens)),
                       "k": K, "segLength": SEG, "sketchSize": SKETCH, "sketchSize_note": W["sketch_note"],
                       "percentageIdentity": PI, "fragments_per_gpu": nF, "resident_batches": nb,
                       "parallelism": "reads sharded, index replicated, RCCL all-gatherv of candidate mappings (libmashmap_hip: mm_allgatherv_mappings_begin/_end, overlapped with the next batch)"
                       if world > 1 else "single GPU", "mean_interval_points_per_fragment": round(P, 1),
                       "l1_candidates_per_gpu": n1, "l2_loci_per_gpu": n2, "candidate_mappings_per_gpu": nmap, "hard_list_fragments": cnts.get("hard", 0),
                       "index_build_s": round(index_s, 2), "rccl": rccl, "reference": ref_summary,
                       "host_synchronisations_last_pass": ctx.pass_stats()[0],
                       "identity_tables": "minimumHits / sketchCutoffs / acceptance from mm_stats.hpp's re-derivation of GSL's binomial and hypergeometric "
                                          "CDFs (GSL is not in the image; pinned by a third derivation: tests/test_host_stats.py, profiles/r13_gsl_boundary_margins.txt)"},
            "passes": dict(passes, note="the %d timed passes went over %d distinct resident batches in rotation: `steady` of them launched everything against the previous pass's buffer "
                                        "sizes and waited for the device once, `redone` outgrew a buffer and were run again the sized way (both inside the timed region)" % (args.steps, nb)),
            "roofline": roofline,
            "kernels": kernels,
        }
        # the side measurements never take the headline with them
        if e2e2 is not None:
            out["e2e"] = e2e2
        if world == 1 and not args.no_host_path and not W.get("assembly"):
            try:
                out["host_path"] = host_path(ctx, W, nreads, ref_lens, step_ms)
            except Exception as e:
                log("[bench] host_path failed:", repr(e)); out["host_path"] = {"error": repr(e)}
        ctx.close(); ctx = None                         # index, parked batches and staging go back to the device before the side measurements
        torch.cuda.empty_cache()
        if want_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(W, ref_np, reads_np, min(args.cpu_sample, nreads))
            except Exception as e:
                log("[bench] cpu_baseline failed:", repr(e)); out["cpu_baseline"] = {"error": repr(e)}
        if want_e2e:
            try:
                t0 = time.time()
                out["e2e"] = e2e_fasta_to_paf(torch, dev, W, ref_np, nreads, max(4, min(128, os.cpu_count() or 1)))
                log("[bench] e2e FASTA -> PAF: %s (%.0f s)" % ({k: out["e2e"].get(k) for k in ("value", "map_s", "device_stage", "error")}, time.time() - t0))
            except Exception as e:
                log("[bench] e2e failed:", repr(e)); out["e2e"] = {"error": repr(e)}
        if is_default and world == 1 and not args.no_north_star:
            # in a process of its own, after this one has let go of its index and reads: whatever happens there, the headline line is printed
            reads_np = ref_np = None
            torch.cuda.empty_cache()
            cmd = [sys.executable, os.path.abspath(__file__), "--north-star-child", "--steps", str(max(1, min(args.steps, args.north_star_steps))),
                   "--warmup", str(min(max(args.warmup, args.batches), 4)), "--cpu-sample", str(args.cpu_sample), "--batches", str(args.batches)] \
                  + (["--no-cpu-baseline"] if args.no_cpu_baseline else []) + (["--no-repeat-rich"] if args.no_repeat_rich else []) + (["--no-configs2"] if args.no_configs2 else [])
            try:
                p = subprocess.run(cmd, stdout=subprocess.PIPE, timeout=1200, env=dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(local))))
                line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
                out["north_star_target"] = json.loads(line[-1]) if p.returncode == 0 and line else {"error": "child exited with %d" % p.returncode}
                if isinstance(out["north_star_target"], dict) and "configs2" in out["north_star_target"]:
                    out["configs2"] = out["north_star_target"].pop("configs2")
            except Exception as e:
                log("[bench] north_star target measurement failed:", repr(e))
                out["north_star_target"] = {"error": repr(e)}
        emit(out)
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
