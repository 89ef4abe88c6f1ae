"""bench_workloads.py -- the synthetic inputs of bench.py (and of the human-scale tests): the BASELINE.json workloads' parameters, reference /
read / assembly generators on the device (torch), the repeat-rich genome model, FASTA writing, and the CPU quota of the process."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

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
