"""tests/humanscale.py -- shared by tests/test_gpu_00_humanscale_start.py and tests/test_gpu_zz_humanscale.py.

Parity at human scale, where other code runs than in the small cases: a 3 Gbp reference indexed on the device (the tagged seed
table above 1 GiB -> k_lookup_l1<.., true>, ~14 GB of open-record lists, candidates located in reference order), mapped through the
`mashmap_hip` command line and compared BYTE FOR BYTE with the PAF of the stock binary (oracle/_ref/mashmap_ref: the reference's own
sources, compiled by oracle/Makefile; it travels to the GPU box like the library does) on the same FASTA files:

  * the north_star target workload: 10 kbp ONT-like reads, defaults (pi 85, segLength 5000), also sharded over two contexts
    (MASHMAP_HIP_DEVICES=0,0);
  * the BASELINE configs[2] shape: assembly contigs vs the reference, --pi 95 -s 10000 -f one-to-one -- a reduced query with an N gap
    (40 x 5 Mbp pieces), and the configuration at FULL size: every contig of the 3 Gbp reference with 1 % substitutions and 1-5 Mbp
    inversions / translocations (bench.make_assembly, what `bench.py --workload configs2` measures), 3 Gbp of query whose records are
    longer than a reader thread's piece of a window (seq_parse.hpp: parseWindowPackedSplit);
  * the BASELINE configs[4] shape: --dense --pi 80, 20 kbp reads at 15-20 % error, the reference as an --rl list of 10 files;
  * the north_star workload once more on a 3 Gbp reference with human-like repeat structure (bench.make_repeat_rich_reference: ~45 % in
    interspersed repeat families of 10^2..10^5 copies at 10-20 % divergence, a satellite array and N gaps per contig) -- the shape
    bench.py reports as north_star_target.repeat_rich: frequent seeds, many interval points and several L1 candidates per fragment,
    fragments on the sketch kernel's hard list.

What is matched: Map::mapQuery end to end (computeMap.hpp:263-413) with the parameters parseCmdArgs.hpp:620-641 derives.
The reference sequence is generated once and shared by the three cases; the stock binary indexes 3 Gbp in ~1.5 minutes per case on
the GPU box's 16 CPUs.  The three runs are started together, as background processes, by the FIRST test of the GPU suite
(test_gpu_00_humanscale_start.py: it makes the files and returns) and are collected by the LAST (test_gpu_zz_humanscale.py), so they
run beside the ~130 other GPU tests instead of in front of them; run alone, the comparison module starts them itself and waits.  MASHMAP_TEST_HUMAN_GBP scales the reference (default 3),
MASHMAP_TEST_HUMAN_READS the read sets."""
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np
import pytest

import mmutil as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_BIN = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
GBP = float(os.environ.get("MASHMAP_TEST_HUMAN_GBP", "3"))
SCALE = float(os.environ.get("MASHMAP_TEST_HUMAN_READS", "1"))
N_CONTIGS, N_FILES = 30, 10                      # 30 contigs of 100 Mbp; the --rl list holds three of them per file
N_READS_NS, N_READS_C4, N_ASM = int(30000 * SCALE), int(6000 * SCALE), max(4, int(40 * SCALE))
N_READS_RR = int(20000 * SCALE)
RR_CONTIGS = 24                                   # 24 x 125 Mbp, as bench.py's north_star workloads
ASM_LEN = 5_000_000


def _threads():
    sys.path.insert(0, ROOT)
    import bench as B
    return str(max(4, min(32, 2 * B.usable_cpus())))


# name -> command line ("@key": a path of the fixture)
CASES = {
    "northstar": ["-r", "@ref", "-q", "@ns"],
    "configs2": ["-r", "@ref", "-q", "@asm", "--pi", "95", "-s", "10000", "-f", "one-to-one"],
    "configs2_full": ["-r", "@ref", "-q", "@asm_full", "--pi", "95", "-s", "10000", "-f", "one-to-one"],
    "configs4": ["--rl", "@rl", "-q", "@c4", "--dense", "--pi", "80"],
    "repeat_rich": ["-r", "@rr_ref", "-q", "@rr"],
}



def start(tmp_path_factory):
    """generator behind the session-scoped `human` fixture (conftest.py)"""
    if not os.path.exists(U.REF_BIN):
        pytest.skip("oracle/_ref/mashmap_ref (the stock binary) is not here: it is built where /root/reference exists and shipped by gpurun")
    assert os.path.exists(HIP_BIN), "mashmap_hip not built (python -c 'import __graft_entry__ as g; g.build()')"
    import torch
    sys.path.insert(0, ROOT)
    import bench as B
    dev = torch.device("cuda", 0)
    td = str(tmp_path_factory.mktemp("human"))
    t0 = time.time()
    clen = int(GBP * 1e9) // N_CONTIGS
    contigs = B.make_reference(torch, dev, N_CONTIGS, clen)
    names = ["chr%d" % i for i in range(N_CONTIGS)]
    ref_fa = os.path.join(td, "ref.fa")
    rl = os.path.join(td, "refs.txt")
    per = N_CONTIGS // N_FILES
    with open(ref_fa, "wb") as whole, open(rl, "w") as lst:
        for fi in range(N_FILES):
            part = os.path.join(td, "ref_part%d.fa" % fi)
            B.write_fasta(part, names[fi * per:(fi + 1) * per], [contigs[i].cpu().numpy() for i in range(fi * per, (fi + 1) * per)])
            lst.write(part + "\n")
            with open(part, "rb") as p:
                shutil.copyfileobj(p, whole, 64 << 20)

    def reads_fasta(path, n, length, err, seed):
        rd = B.make_reads(torch, dev, contigs, n, length, err, seed=seed).cpu().numpy().reshape(n, length)
        B.write_fasta(path, ["read%d" % i for i in range(n)], list(rd), width=length)

    ns_fa, c4_fa, asm_fa = (os.path.join(td, x) for x in ("reads_ns.fa", "reads_c4.fa", "asm.fa"))
    reads_fasta(ns_fa, N_READS_NS, 10000, (0.10, 0.10), 1000)
    reads_fasta(c4_fa, N_READS_C4, 20000, (0.15, 0.20), 2000)
    # "assembly": pieces of the reference with 1 % substitutions, every third one reverse-complemented, one with a gap of Ns
    g = torch.Generator(device=dev); g.manual_seed(77)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    asm = []
    for i in range(N_ASM):
        ci = (i * 7) % N_CONTIGS
        st = int(torch.randint(0, clen - ASM_LEN, (1,), generator=g, device=dev))
        c = contigs[ci][st:st + ASM_LEN].clone()
        m = torch.rand(ASM_LEN, generator=g, device=dev) < 0.01
        c = torch.where(m, lut[torch.randint(0, 4, (ASM_LEN,), generator=g, device=dev)], c)
        if i % 3 == 2:
            c = comp[c.flip(0).long()]
        if i == 1:
            c[ASM_LEN // 2:ASM_LEN // 2 + 50000] = ord("N")
        asm.append(c.cpu().numpy())
    B.write_fasta(asm_fa, ["ctg%d" % i for i in range(N_ASM)], asm)
    asm_full_fa = os.path.join(td, "asm_full.fa")
    full_asm = B.make_assembly(torch, dev, contigs, 0.01, seed=2021)
    B.write_fasta(asm_full_fa, ["asm%d" % i for i in range(len(full_asm))], [c.cpu().numpy() for c in full_asm])
    del contigs, asm, full_asm
    torch.cuda.empty_cache()
    # the repeat-rich reference and reads drawn from it
    rr_ref, rr_fa = os.path.join(td, "ref_rr.fa"), os.path.join(td, "reads_rr.fa")
    contigs, rr_summary = B.make_repeat_rich_reference(torch, dev, RR_CONTIGS, int(GBP * 1e9) // RR_CONTIGS)
    B.write_fasta(rr_ref, ["chr%d" % i for i in range(RR_CONTIGS)], [c.cpu().numpy() for c in contigs])
    reads_fasta(rr_fa, N_READS_RR, 10000, (0.10, 0.10), 3000)
    del contigs
    torch.cuda.empty_cache()
    print("\n[human scale] %.2f Gbp reference in %d contigs (+ %d --rl files), %d + %d reads, %d assembly contigs written in %.0f s"
          % (GBP, N_CONTIGS, N_FILES, N_READS_NS, N_READS_C4, N_ASM, time.time() - t0), flush=True)
    H = dict(td=td, ref=ref_fa, rl=rl, ns=ns_fa, c4=c4_fa, asm=asm_fa, asm_full=asm_full_fa, rr_ref=rr_ref, rr=rr_fa, rr_summary=rr_summary, threads=_threads())
    # the four runs of the stock binary start now and share the host's CPUs (its index build is single-threaded for most of its
    # ~90 s: hash-map insertions of 0.36 G records); the GPU runs of the tests below happen meanwhile
    H["stock"] = {}
    for name, args in CASES.items():
        full = [a if not a.startswith("@") else H[a[1:]] for a in args] + ["-t", H["threads"]]
        out = os.path.join(td, name + ".ref.paf")
        log = open(os.path.join(td, name + ".ref.log"), "w")
        H["stock"][name] = (subprocess.Popen([U.REF_BIN] + full + ["-o", out], stdout=log, stderr=subprocess.STDOUT), out, log, time.time(), full)
    yield H
    for p, _, log, _, _ in H["stock"].values():
        if p.poll() is None:
            p.kill()
        log.close()
    shutil.rmtree(td, ignore_errors=True)


def _run(exe, args, out, env=None):
    t0 = time.time()
    e = dict(os.environ, MASHMAP_HIP_TIMING="1")
    if env:
        e.update(env)
    p = subprocess.run([exe] + args + ["-o", out], capture_output=True, text=True, env=e)
    assert p.returncode == 0, "%s %s\n%s" % (exe, " ".join(args), p.stderr[-3000:])
    tm = {k.split()[0]: float(v) for k, v in re.findall(r"time spent (computing the reference index|mapping the query)\s*:\s*([0-9.eE+-]+)", p.stderr)}
    return open(out, "rb").read(), p.stderr, time.time() - t0, tm


def _diff(a, b):
    la, lb = a.decode().splitlines(), b.decode().splitlines()
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d:\n  got %s\n  exp %s\n(%d vs %d lines)" % (i, x, y, len(la), len(lb))
    return "%d vs %d lines; first extra: %s" % (len(la), len(lb), (la[len(lb):] or lb[len(la):])[:1])


def _tagged(stderr):
    m = re.search(r"index layout: seed table (\d+) slots \((\d+) MiB\), tagged=(\d)", stderr)
    assert m, "no index layout line in the MASHMAP_HIP_TIMING log:\n" + stderr[-1500:]
    return int(m.group(3)) >= 1, int(m.group(2))


def case(human, name, min_lines, expect_tagged, sharded=False):
    td = human["td"]
    proc, ref_out, log, t_start, full = human["stock"][name]
    got, err, wall_h, tm_h = _run(HIP_BIN, full, os.path.join(td, name + ".hip.paf"))
    tagged, mib = _tagged(err)
    if GBP >= 2.5 and expect_tagged:
        assert tagged and mib > 1024, "the tagged seed table (k_lookup_l1<.., true>) is not in use at this scale: %d MiB, tagged=%s" % (mib, tagged)
    rc = proc.wait()
    wall_r = time.time() - t_start
    log.flush()
    ref_err = open(log.name).read()
    assert rc == 0, "stock binary %s\n%s" % (" ".join(full), ref_err[-3000:])
    tm_r = {k.split()[0]: float(v) for k, v in re.findall(r"time spent (computing the reference index|mapping the query)\s*:\s*([0-9.eE+-]+)", ref_err)}
    exp = open(ref_out, "rb").read()
    print("\n[human scale] %s: mashmap_hip %.1f s %s | stock binary done %.1f s after the module's start %s | %d PAF lines, seed table %d MiB tagged=%s"
          % (name, wall_h, tm_h, wall_r, tm_r, exp.count(b"\n"), mib, tagged), flush=True)
    assert exp.count(b"\n") >= min_lines, "the stock binary mapped only %d lines" % exp.count(b"\n")
    assert got == exp, _diff(got, exp)
    if sharded:
        got2, err2, wall2, _ = _run(HIP_BIN, full, os.path.join(td, name + ".hip2.paf"), env={"MASHMAP_HIP_DEVICES": "0,0"})
        print("[human scale] %s sharded over two contexts: %.1f s" % (name, wall2), flush=True)
        assert got2 == exp, "MASHMAP_HIP_DEVICES=0,0: " + _diff(got2, exp)


