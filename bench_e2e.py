"""bench_e2e.py -- the side measurements of bench.py that run on the host or through the command line: the stock binary as `cpu_baseline`
(oracle/_ref/mashmap_ref: the checker's build of the reference, timed outside bench.py's timed region), the host stage of skch::Map
(`host_path`), and FASTA -> PAF through `mashmap_hip` with its stage log (`e2e` for configs[1], `e2e_assembly` for configs[2])."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

from bench_workloads import ROOT, log, make_assembly, make_reads, usable_cpus, write_fasta

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
