#!/usr/bin/env python3
"""Opcode histogram of the sketch kernel's inner loop next to the hash-only kernel's (the floor it is measured against), per k-mer position.

    python scripts/isa_histogram.py [--k 19 --strip 20] [--asm FILE] > profiles/rNN_sketch_opcode_histogram.txt

Compiles mashmap_amd/csrc/mm_sketch.hip to gfx950 assembly (hipcc -S --cuda-device-only, ~2 minutes, no GPU needed; --asm reuses a file),
finds in k_hash_only<K,STRIP> and k_sketch_fast<K,STRIP> the loop that walks a thread's strip of STRIP positions -- the largest loop body of
either kernel: the compiler unrolls the strip completely, so one iteration = STRIP positions --, and counts its instructions by opcode.
Per position = count / STRIP.  The issue cost column is what scripts/probes/valu_rate.hip measured for that opcode on the part at 8 waves per
SIMD (profiles/r02_valu_rate.txt; opcodes it did not time are priced by their encoding class: VOP3 ~4.2, VOP1/VOP2 ~2.3 cycles)."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# cycles per wave64 instruction per SIMD at 8 waves/SIMD (profiles/r02_valu_rate.txt); default by class below
COST = {"v_add_u32": 2.35, "v_xor_b32": 2.27, "v_mul_lo_u32": 4.16, "v_mul_hi_u32": 4.11, "v_alignbit_b32": 4.15, "v_add3_u32": 4.16, "v_perm_b32": 4.16,
        "v_lshl_add_u32": 4.14, "v_bfe_u32": 4.18, "v_lshrrev_b32": 2.3, "v_lshlrev_b32": 2.3, "v_mov_b32": 2.3, "v_mad_u64_u32": 4.3, "v_lshl_add_u64": 4.3,
        "v_lshlrev_b64": 4.2, "v_lshrrev_b64": 4.2, "v_cmp_lt_u64": 4.3}
VOP2ISH = ("v_add_", "v_sub", "v_xor_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_not_b32", "v_addc", "v_subb",
           "v_min_", "v_max_", "v_cndmask", "v_accvgpr")


def cost(op):
    base = op[:-4] if op.endswith(("_e32", "_e64")) else op
    if base in COST:
        return COST[base]
    if op.endswith("_e64") or not base.startswith(VOP2ISH):
        return 4.2
    return 2.3


def kernel_text(asm, mangled_prefix):
    start = next((i for i, l in enumerate(asm) if l.startswith(mangled_prefix) and ":" in l), None)
    assert start is not None, "kernel %s not in the assembly" % mangled_prefix
    end = start
    while end < len(asm) and "s_endpgm" not in asm[end]:
        end += 1
    return asm[start:end + 1]


def largest_loop(lines):
    """(first, last) line indices of the longest label .. backward-branch span"""
    labels = {}
    best = (0, 0)
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
            continue
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and i - labels[m.group(1)] > best[1] - best[0]:
            best = (labels[m.group(1)], i)
    return best


def histogram(lines):
    h = collections.Counter()
    for l in lines:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        h[l.split()[0]] += 1
    return h


def klass(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=19)
    ap.add_argument("--strip", type=int, default=20)
    ap.add_argument("--asm", default="")
    a = ap.parse_args()
    asm_path = a.asm
    if not asm_path:
        asm_path = os.path.join(tempfile.gettempdir(), "mm_sketch_gfx950.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                               "-o", asm_path, os.path.join(ROOT, "mashmap_amd", "csrc", "mm_sketch.hip")], stderr=subprocess.DEVNULL)
    asm = open(asm_path).read().splitlines()
    kern = {"k_hash_only": "_Z11k_hash_onlyILi%dELi%dEE" % (a.k, a.strip), "k_sketch_fast": "_Z13k_sketch_fastILi%dELi%dEE" % (a.k, a.strip)}
    hist, meta = {}, {}
    for name, pre in kern.items():
        txt = kernel_text(asm, pre)
        lo, hi = largest_loop(txt)
        hist[name] = histogram(txt[lo:hi + 1])
        meta[name] = (len(txt), hi - lo + 1, sum(hist[name].values()))
    S = float(a.strip)
    print("# opcode histogram of the strip loop, k = %d, %d positions per thread and iteration (the loop is unrolled over the strip)" % (a.k, a.strip))
    for name in kern:
        print("# %-14s kernel %6d lines, strip loop %6d lines, %6d instructions = %.1f per position" % (name, meta[name][0], meta[name][1], meta[name][2], meta[name][2] / S))
    print("#\n# per position:  %-22s %10s %10s %8s   %s" % ("opcode", "hash_only", "sketch", "extra", "cycles/instr (8 waves/SIMD)"))
    ops = sorted(set(hist["k_hash_only"]) | set(hist["k_sketch_fast"]), key=lambda o: (klass(o), -(hist["k_sketch_fast"][o] + hist["k_hash_only"][o])))
    tot = {n: collections.Counter() for n in kern}
    cyc = {n: 0.0 for n in kern}
    for o in ops:
        hcount, scount = hist["k_hash_only"][o] / S, hist["k_sketch_fast"][o] / S
        c = cost(o) if klass(o) == "VALU" else 0.0
        print("%-6s %-32s %10.2f %10.2f %+8.2f   %s" % (klass(o), o, hcount, scount, scount - hcount, ("%.2f" % c) if c else "-"))
        for n, v in (("k_hash_only", hcount), ("k_sketch_fast", scount)):
            tot[n][klass(o)] += v; cyc[n] += v * c
    print("#\n# per position, by class:")
    for kl in ("VALU", "SALU", "LDS", "VMEM", "wait", "other"):
        print("# %-6s %10.2f %10.2f %+8.2f" % (kl, tot["k_hash_only"][kl], tot["k_sketch_fast"][kl], tot["k_sketch_fast"][kl] - tot["k_hash_only"][kl]))
    print("# VALU issue cycles per position (sum of count x measured cost): hash_only %.1f, sketch %.1f  => the strip loop alone bounds sketch_kernel_frac at %.3f"
          % (cyc["k_hash_only"], cyc["k_sketch_fast"], cyc["k_hash_only"] / cyc["k_sketch_fast"]))
    print("# (what the loop does not contain: the survivor path -- queue drain, table insert, rank scan, output -- paid per fragment outside it)")


if __name__ == "__main__":
    sys.exit(main())
