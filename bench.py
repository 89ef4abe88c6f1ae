#!/usr/bin/env python3
"""bench.py -- query Gbp/s of the sketch + L1/L2 hot path on N MI355X (one process per GPU).

  python bench.py --gpus N --steps K --warmup W [--workload configs1|configs3|configs4]

With N > 1 and no torch.distributed environment the script starts its own N ranks (torch.distributed.run, 127.0.0.1); started by
a launcher it checks that WORLD_SIZE == N.  It never prints an `n_gpus` other than the N it was asked for.

A "step" = one pass of the hot path (sketch -> seed lookup -> L1 -> L2 slide -> doL2Mapping's best-first selection) over the
resident batch; inputs (2-bit packed bases + N mask) are already in HBM when the timed region starts.  For N > 1 every rank maps
its own reads (weak scaling: `reads` per GPU, index replicated) and the step ends with the RCCL all-gatherv of the candidate
mappings (mm_allgatherv_mappings, mashmap_amd/csrc/mm_comm.hip) -- the product's own exchange step, not a Python stand-in.

Workloads (BASELINE.json `configs`; the default is configs[1], the configuration the metric is quoted on):
  configs1  1 M x 10 kbp reads (10 % ONT-like error) vs 100 Mbp, pi 85, segLength 5000, sketchSize 130
  configs3  per-GPU share of configs[3]: 1.25 M x 15 kbp reads vs 3 Gbp (24 x 125 Mbp), sketchSize 310 (the stock binary's value:
            its int32 referenceSize overflows for a 3 GB file; 220 mathematically -- SURVEY App. C)
  northstar the north_star target sentence: 1 M x 10 kbp reads, pi 85, against the 3 Gbp index (sketchSize 310 as for configs3)
  configs4  per-GPU share of configs[4]: 625 k x 20 kbp reads at 15-20 % error vs 10 x 300 Mbp (the --rl list shares one seqId
            space, winSketch.hpp:174-214), --dense --pi 80 => sketchSize 498
--reads / --ref-contigs / --ref-contig-len scale a workload down; the JSON line names what actually ran.

The default run (configs1, nothing scaled, one GPU) also measures the north_star target sentence itself -- 1 M x 10 kbp reads at pi 85
against the human-scale (3 Gbp) index -- after the headline measurement and attaches it as the extra key `north_star_target`
(stock segLength 5000, and a segLength 10000 variant: the sentence says "10 kbp segments"); `value` / `config` / `roofline` stay the
configs[1] figures.  --no-north-star skips it.
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
                return {"value": n_sample * read_len / tmap / 1e9, "unit": "Gbp/s", "cores": nt, "kind": "reference",
                        "sample": desc + "; mashmap_ref (built from the reference sources) best of -t 8/32/64 = -t %d of %d host cores, "
                                         "'time spent mapping the query' (includes its single-threaded FASTA reader); this process may use %d CPUs at once (affinity / container quota)" % (nt, ncores, usable_cpus()),
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


def pmc_entry(workload_key, kernel="k_sketch_fast"):
    """counters of one kernel from the committed PMC passes of this workload (profiles/pmc_traffic.json: per workload, per kernel,
    per launch), and where they came from -- they are NOT measured in the bench run itself"""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        wl = json.load(open(pmc)).get(workload_key)
        if not wl or kernel not in wl:
            return None, None
        return wl[kernel], "profiles/pmc_traffic.json [%s] (rocprofv3 --pmc passes %s of this workload; not measured in this run)" % (workload_key, wl.get("_source", "?"))
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
        ent, src = pmc_entry(workload_key, kern) if pmc_ok else (None, None)
        out.append({"kernel": kern, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                    "avg_ms_per_pass": round(avg, 3), "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes": what, "binds": binds,
                    "traffic": ent.get("hbm_bytes_per_launch") if ent else None,
                    "valu_wave_instructions_per_launch": ent.get("SQ_INSTS_VALU") if ent else None, "source": src})
    return out


def roofline_block(ctx, capi, W, workload_key, nF, prof, step_ms, pmc_ok, mean_points=0.0):
    """roofline of the dominant kernel (k_sketch_fast): algorithmic bytes per fragment = L/4 packed bases in + 24 B per sketch entry out
    (SURVEY section 8d), divided by its average HIP-event duration in the timed region; the integer yardstick; VALU issue peaks"""
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
    traffic = valu = source = None
    if pmc_ok:
        ent, source = pmc_entry(workload_key)
        if ent:
            traffic = ent.get("hbm_bytes_per_launch")
            ninst = ent.get("SQ_INSTS_VALU")
            if ninst and sk_avg > 0:
                per_s = ninst / (sk_avg * 1e-3)
                mix = 3.35                          # cycles per wave-instruction of this kernel's VOP3/VOP2 mix (profiles/r02_valu_rate.txt, r02_sketch_instruction_mix.txt)
                valu = {"wave_instructions_per_launch": ninst,
                        "util_vs_2_cycles_per_instr": round(per_s / (SIMDS * CLOCK_HZ / 2.0), 3),
                        "util_vs_measured_mix": round(per_s / (SIMDS * CLOCK_HZ / mix), 3),
                        "util_vs_4_cycles_per_instr": round(per_s / (SIMDS * CLOCK_HZ / 4.0), 3),
                        "model": "1024 SIMDs x 2.4 GHz; three issue peaks: 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md nominal, SIMD-32), "
                                 "%.2f cycles (this kernel's mix of VOP3 integer ops at ~4.2 and VOP2 ops at ~2.3 cycles measured by "
                                 "scripts/probes/valu_rate.hip, output in profiles/), 4 cycles (every instruction at the VOP3 rate)" % mix}
    kernels = kernel_rooflines(ctx, W, workload_key, nF, prof, mean_points, pmc_ok)
    return {"bound": "hbm", "kernel": "k_sketch_fast", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": kernels,
            "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": source,
            "algorithmic_bytes_per_fragment": frag_bytes, "avg_launch_ms": round(sk_avg, 3),
            "algorithmic_bytes_per_launch": frag_bytes * nF,
            "note": "integer-issue bound kernel (2 x MurmurHash3_x64_128 per base); the HBM fraction is small by construction -- "
                    "`int` (hash-only yardstick) and `valu` (issue peaks) are the rooflines that bind, DESIGN.md section 3",
            "int": integer, "valu": valu}


def timed_passes(ctx, warmup, steps):
    """W untimed passes, then K timed ones bracketed by a stream synchronisation; returns (seconds, per-kernel HIP-event times)"""
    for _ in range(warmup):
        ctx.map()
    ctx.profile(True); ctx.profile_read(reset=True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.map()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    return dt, prof


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
    return {"value": round(n_sample * L / tm["mapping the query"] / 1e9, 4), "unit": "Gbp/s", "cores": nt, "kind": "reference",
            "index_build_s": round(tm.get("computing the reference index", 0.0), 1), "wall_s": round(wall, 1), "paf_lines": lines,
            "sample": "%d of the target's reads (%.0f Mbp) vs the same %.0f Mbp reference written as FASTA; mashmap_ref (the reference's sources, GSL stand-in) with its "
                      "defaults, -t %d: %d host hardware threads, of which this process may use %d CPUs at once (container quota); 'time spent mapping the query' includes its "
                      "single-threaded FASTA reader" % (n_sample, n_sample * L / 1e6, sum(len(a) for a in ref_np) / 1e6, nt, os.cpu_count() or 1, usable_cpus())}


def north_star_target(torch, dev, capi, local, warmup, steps, cpu_reads=0):
    """BASELINE.json's north_star target sentence on one GPU: 1 M x 10 kbp ONT-like reads, pi 85, against a human-scale index (3 Gbp,
    24 x 125 Mbp), device-resident packed bases in -> candidate mappings out.  Two variants on the same data: the stock command line
    (segLength 5000, the metric's "s=5000": two fragments per read) and segLength 10000 ("10 kbp segments": one fragment per read);
    sketchSize 310 = what the stock binary derives for a 3 GB reference file at either segment length."""
    base = dict(WORKLOADS["northstar"])
    t0 = time.time()
    contigs = make_reference(torch, dev, base["ref_contigs"], base["ref_contig_len"])
    ref_np = contiguous_views(torch, contigs)
    reads_t = make_reads(torch, dev, contigs, base["reads"], base["read_len"], base["err"], seed=1000)
    torch.cuda.synchronize()
    del contigs
    torch.cuda.empty_cache()
    gen_s = time.time() - t0
    nreads, READ_LEN = base["reads"], base["read_len"]
    offs = np.arange(nreads + 1, dtype=np.int64) * READ_LEN
    res = {}
    for key, seg in (("segLength5000", 5000), ("segLength10000", 10000)):
        W = dict(base, seg=seg)
        ctx = capi.Context(k=W["k"], segLength=seg, sketchSize=W["sketch"], flags=capi.MM_FLAG_HG_FILTER, device=local)
        t0 = time.time()
        ctx.index_build(ref_np, kmerPct=0.001)
        index_s = time.time() - t0
        ctx.set_tables_default(W["pi"])
        nF = ctx.reads_upload_device(reads_t.data_ptr(), reads_t.numel(), offs, seqCounterBase=0)
        dt, prof = timed_passes(ctx, warmup, steps)
        step_ms = dt / steps * 1e3
        n1, n2 = ctx.result_counts()
        stats, _, _ = ctx.results()
        nmap = len(ctx.mappings())
        wl_key = "northstar" if seg == 5000 else "northstar_seg10000"
        res[key] = {
            "value": round(nreads * READ_LEN * steps / dt / 1e9, 4), "unit": "Gbp/s", "ms_per_step": round(step_ms, 3), "steps": steps, "warmup": warmup,
            "kernels": {k: {"ms_per_step": v[0] / steps, "launches_per_step": v[1] / steps} for k, v in prof.items() if v[1]},
            "roofline": roofline_block(ctx, capi, W, wl_key, nF, prof, step_ms, True, float(stats["nPoints"].mean())),
            "index_build_s": round(index_s, 2),
            "workload": "%d x %d bp reads (10%% ONT-like error) vs %.0f Mbp synthetic reference (%d contigs), k %d, segLength %d, sketchSize %d, pi %.2f: "
                        "%d fragments, %.1f interval points per fragment, %d L1 candidates, %d L2 loci, %d candidate mappings"
                        % (nreads, READ_LEN, sum(len(a) for a in ref_np) / 1e6, len(ref_np), W["k"], seg, W["sketch"], W["pi"], nF,
                           float(stats["nPoints"].mean()), n1, n2, nmap)}
        log("[north_star] %s: %.1f Gbp/s, %.1f ms per pass, index %.1f s" % (key, res[key]["value"], step_ms, index_s))
        ctx.close()
        del ctx
    out = dict(res["segLength5000"])
    if cpu_reads > 0:
        try:
            out["cpu_baseline"] = human_scale_cpu_baseline(base, ref_np, reads_t, cpu_reads)
        except Exception as e:
            log("[north_star] cpu_baseline failed:", repr(e)); out["cpu_baseline"] = {"error": repr(e)}
    out["what"] = ("BASELINE.json north_star target (>= 50 query Gbp/s sketch+map on 1 x MI355X, 10 kbp reads at pi 85 against a human-scale index), measured in "
                   "this run behind the headline configuration; inputs resident in HBM, same timed region as `value`")
    out["target_gbps"] = 50.0
    out["sketchSize_note"] = base["sketch_note"]
    out["synthetic_data_s"] = round(gen_s, 2)
    out["segLength_10000"] = res["segLength10000"]
    return out


class StubContext:
    """CPU stand-in used ONLY by tests/test_bench_spawn.py (--stub): lets the launcher / rank plumbing of this script run where there
    is no GPU.  It maps nothing; a line produced with it says "data": "stub"."""

    def __init__(self, rank):
        self.rank = rank

    def map(self):
        time.sleep(0.002)

    def allgatherv(self, dist):
        from mashmap_amd import capi, shard
        mine = np.zeros(3 + self.rank, dtype=capi.MAPPING_DT); mine["querySeqId"] = self.rank
        got, counts = shard.allgatherv_mappings(mine, dist)
        assert len(got) == sum(counts) and (got["querySeqId"] == np.repeat(np.arange(len(counts)), counts)).all()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="configs1")
    ap.add_argument("--reads", type=int, default=int(os.environ.get("MM_BENCH_READS", 0)), help="reads per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--no-north-star", action="store_true", help="default run only: skip the north_star target measurement (3 Gbp index) behind the headline one")
    ap.add_argument("--north-star-steps", type=int, default=5, help="timed passes of each north_star variant (at most --steps)")
    ap.add_argument("--ref-contigs", type=int, default=0, help="contigs of the synthetic reference (default: the workload's)")
    ap.add_argument("--ref-contig-len", type=int, default=0)
    ap.add_argument("--kmer", type=int, default=0, help="k-mer size (default: the reference's 19; other sizes are not the BASELINE configuration)")
    ap.add_argument("--seg", type=int, default=0, help="segLength (default: the workload's; `--workload northstar --seg 10000` is the north_star sentence's 10 kbp segments, "
                                                       "sketchSize unchanged, as north_star_target.segLength_10000 measures it)")
    ap.add_argument("--cpu-sample", type=int, default=30000)
    ap.add_argument("--sync-exchange", action="store_true", help="N>1: all-gatherv on the compute stream instead of overlapped with the next batch")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--north-star-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.north_star_child:                           # the default run's second measurement (see north_star_target), one GPU
        import torch
        from mashmap_amd import capi
        torch.cuda.set_device(0)
        print(json.dumps(north_star_target(torch, torch.device("cuda", 0), capi, 0, args.warmup, args.steps, 0 if args.no_cpu_baseline else args.cpu_sample)), flush=True)
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
    wl_key = args.workload
    if args.seg and args.seg != W["seg"]:
        W["seg"] = args.seg; W["label"] += ", segLength %d" % args.seg
        wl_key = "%s_seg%d" % (args.workload, args.seg)
    is_default = args.workload == "configs1" and not scaled and wl_key == args.workload

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.stub else "nccl", rank=rank, world_size=world)

    if args.stub:
        ctx = StubContext(rank)
        for _ in range(args.warmup):
            ctx.map(); world > 1 and ctx.allgatherv(dist)
        if world > 1: dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.map(); world > 1 and ctx.allgatherv(dist)
        if world > 1: dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tm = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(tm, op=dist.ReduceOp.MAX); dt = float(tm.item())
        if rank == 0:
            print(json.dumps({"metric": "query Gbp/s sketch+L1/L2 map (pi=85, s=5000)", "value": 0.0, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "u64", "data": "stub", "config": {"workload": "STUB: no kernels ran (launcher test)"}}), flush=True)
        if world > 1:
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
    contigs = make_reference(torch, dev, W["ref_contigs"], W["ref_contig_len"])
    ref_np = contiguous_views(torch, contigs)
    reads_t = make_reads(torch, dev, contigs, nreads, READ_LEN, W["err"], seed=1000 + rank)
    torch.cuda.synchronize()
    log("[rank %d] synthetic data: %.1f s" % (rank, time.time() - t0))
    del contigs
    torch.cuda.empty_cache()

    ctx = capi.Context(k=K, segLength=SEG, sketchSize=SKETCH, flags=capi.MM_FLAG_HG_FILTER, device=local)
    t0 = time.time()
    ctx.index_build(ref_np, kmerPct=0.001)
    index_s = time.time() - t0
    t0 = time.time()
    ctx.set_tables_default(PI)
    log("[rank %d] index build: %.1f s; integer tables: %.2f s" % (rank, index_s, time.time() - t0))
    offs = np.arange(nreads + 1, dtype=np.int64) * READ_LEN
    nF = ctx.reads_upload_device(reads_t.data_ptr(), reads_t.numel(), offs, seqCounterBase=rank * nreads)
    # the CPU leg indexes the reference with the stock binary: minutes beyond a few hundred Mbp, so it rides on the default workload only
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and sum(len(a) for a in ref_np) <= 400e6
    reads_np = reads_t[:min(nreads, args.cpu_sample) * READ_LEN].cpu().numpy() if want_cpu else None
    ref_lens = [len(a) for a in ref_np]
    if not want_cpu:
        ref_np = None
    del reads_t
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

    inflight = [False]

    def step():
        ctx.map()
        if world > 1:                                   # all-gatherv of the candidate mappings over RCCL/xGMI: the exchange of batch i
            if args.sync_exchange:                      # runs on the library's exchange stream under the kernels of batch i+1
                ctx.allgatherv_mappings()               # (--sync-exchange: on the compute stream, the step waits for it)
                return
            if inflight[0]:
                ctx.allgatherv_mappings_end()
            ctx.allgatherv_mappings_begin()
            inflight[0] = True

    def fence():                                        # the last exchange is waited for INSIDE the timed region
        if inflight[0]:
            ctx.allgatherv_mappings_end()
            inflight[0] = False
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile(True); ctx.profile_read(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    n1, n2 = ctx.result_counts()

    if rank == 0:
        stats, _, _ = ctx.results()
        nmap = len(ctx.mappings())
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
            "config": {"workload": "%s%s: %d x %d bp reads/GPU (%s ONT-like error) vs %.0f Mbp synthetic reference (%d contigs)"
                                   % (W["label"], " SCALED (%s)" % ", ".join(scaled) if scaled else "", nreads, READ_LEN,
                                      "%.0f%%" % (W["err"][0] * 100) if W["err"][0] == W["err"][1] else "%.0f-%.0f%%" % (W["err"][0] * 100, W["err"][1] * 100),
                                      ref_mbp, len(ref_lens)),
                       "k": K, "segLength": SEG, "sketchSize": SKETCH, "sketchSize_note": W["sketch_note"],
                       "percentageIdentity": PI, "fragments_per_gpu": nF,
                       "parallelism": "reads sharded, index replicated, RCCL all-gatherv of candidate mappings (libmashmap_hip: mm_allgatherv_mappings_begin/_end, overlapped with the next batch)"
                       if world > 1 else "single GPU", "mean_interval_points_per_fragment": round(P, 1),
                       "l1_candidates_per_gpu": n1, "l2_loci_per_gpu": n2, "candidate_mappings_per_gpu": nmap,
                       "index_build_s": round(index_s, 2), "rccl": rccl,
                       "host_synchronisations_per_pass": ctx.pass_stats()[0],
                       "identity_tables": "minimumHits / sketchCutoffs / acceptance from mm_stats.hpp's re-derivation of GSL's binomial and hypergeometric "
                                          "CDFs (GSL is not in the image; SURVEY section 8c: the one unpinned boundary)"},
            "roofline": roofline,
            "kernels": kernels,
        }
        # the two side measurements never take the headline with them
        if world == 1 and not args.no_host_path:
            try:
                out["host_path"] = host_path(ctx, W, nreads, ref_lens, step_ms)
            except Exception as e:
                log("[bench] host_path failed:", repr(e)); out["host_path"] = {"error": repr(e)}
        if want_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(W, ref_np, reads_np, min(args.cpu_sample, nreads))
            except Exception as e:
                log("[bench] cpu_baseline failed:", repr(e)); out["cpu_baseline"] = {"error": repr(e)}
        if is_default and world == 1 and not args.no_north_star:
            # in a process of its own, after this one has let go of its index and reads: whatever happens there, the headline line is printed
            ctx.close(); ctx = None
            del reads_np, ref_np
            torch.cuda.empty_cache()
            cmd = [sys.executable, os.path.abspath(__file__), "--north-star-child", "--steps", str(max(1, min(args.steps, args.north_star_steps))),
                   "--warmup", str(min(args.warmup, 2)), "--cpu-sample", str(args.cpu_sample)] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
            try:
                p = subprocess.run(cmd, stdout=subprocess.PIPE, timeout=900, env=dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(local))))
                line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
                out["north_star_target"] = json.loads(line[-1]) if p.returncode == 0 and line else {"error": "child exited with %d" % p.returncode}
            except Exception as e:
                log("[bench] north_star target measurement failed:", repr(e))
                out["north_star_target"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
