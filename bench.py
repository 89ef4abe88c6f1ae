#!/usr/bin/env python3
"""bench.py -- query Gbp/s of the sketch + L1/L2 hot path on N MI355X (one process per GPU).

  python bench.py --gpus N --steps K --warmup W [--workload configs1|configs2|configs3|configs4|northstar] [--batches B]

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
  configs2  the 3 Gbp assembly (every reference contig with 1 % substitutions and 1-5 Mbp inversions / translocations) vs the 3 Gbp
            reference, pi 95, segLength 10000, sketchSize 40 (-J 40), -f one-to-one: resident passes + FASTA -> PAF (--stock: the stock
            binary on the same files, PAF bytes compared)
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
--no-e2e / --no-north-star / --no-configs2 skip them.

What rank 0 prints LAST is one JSON object of about 4 KB (compact_line: the contract's keys, `roofline`, `cpu_baseline`, one small numeric
object per side measurement); the full record of the run goes to profiles/bench_last_full.json.  The synthetic inputs live in
bench_workloads.py, the host / command-line side measurements in bench_e2e.py; this file holds the timed loop, the roofline and the line.
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

from bench_workloads import *      # noqa: F401,F403  (WORKLOADS, the generators, write_fasta, usable_cpus, log: tests and scripts reach them as bench.X)
from bench_workloads import WORKLOADS, contiguous_views, log, make_assembly, make_reads, make_reference, make_repeat_rich_reference, usable_cpus, write_fasta  # noqa: F401
from bench_e2e import cpu_baseline, e2e_assembly, e2e_fasta_to_paf, host_path, human_scale_cpu_baseline, run_cli_staged, stage_summary  # noqa: F401


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
