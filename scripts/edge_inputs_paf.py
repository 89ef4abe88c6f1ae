"""Awkward input files through both command lines (mashmap_hip and the stock binary built from the reference sources): return code and
PAF bytes must agree.  Empty files, reads shorter than k / than a segment, N-only and lower-case reads, multi-line and CRLF FASTA,
FASTQ, records without sequence, a reference contig shorter than a window, duplicate names, a missing trailing newline."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mmutil as U

HIP = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
td = tempfile.mkdtemp()
g = [U.random_dna(31 + i, n) for i, n in enumerate((120000, 60000))]
S = lambda a: a.tobytes().decode()
def fasta(recs, width=0, eol="\n", last_eol=True):
    out = []
    for n, s in recs:
        out.append(">" + n + eol)
        if width: out += [s[i:i + width] + eol for i in range(0, len(s), width)]
        else: out.append(s + eol)
    t = "".join(out)
    return t if last_eol else t.rstrip("\r\n")
ref = fasta([("chr0", S(g[0])), ("chr1", S(g[1]))], 70)
reads = [("r%d" % i, S(U.mutate(g[i % 2][o:o + 12000], 50 + i, 0.06))) for i, o in enumerate((1000, 20000, 40000, 7000))]
cases = {
  "plain":            (ref, fasta(reads)),
  "empty_query":      (ref, ""),
  "only_short_reads": (ref, fasta([("s1", "ACGTACGTAC"), ("s2", S(g[0][:3000]))])),
  "shorter_than_k":   (ref, fasta(reads[:1] + [("tiny", "ACGT")] + reads[1:2])),
  "n_only_read":      (ref, fasta(reads[:1] + [("allN", "N" * 9000)] + reads[1:])),
  "lower_case":       (ref.lower().replace(">chr", ">chr"), fasta([(n, s.lower()) for n, s in reads])),
  "multi_line_60":    (ref, fasta(reads, 60)),
  "crlf":             (fasta([("chr0", S(g[0])), ("chr1", S(g[1]))], 70, "\r\n"), fasta(reads, 80, "\r\n")),
  "no_final_newline": (ref, fasta(reads, 0, "\n", False)),
  "empty_record":     (ref, fasta(reads[:2] + [("nothing", "")] + reads[2:])),
  "name_with_spaces": (ref, fasta([(n + " some description here", s) for n, s in reads])),
  "duplicate_names":  (ref, fasta([("same", s) for _, s in reads])),
  "tiny_contig":      (fasta([("chr0", S(g[0])), ("bit", S(g[1][:300])), ("chr1", S(g[1]))], 70), fasta(reads)),
  "fastq":            (ref, "".join("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, s in reads)),
  "iupac":            (ref, fasta([(n, s[:5000] + "RYKMSWBDHVN" * 3 + s[5033:]) for n, s in reads])),
  "blank_lines":      (ref, fasta(reads, 100).replace("\n>", "\n\n>")),
}
bad = 0
for name, (rtxt, qtxt) in cases.items():
    rf, qf = os.path.join(td, name + "_ref.fa"), os.path.join(td, name + ("_q.fq" if name == "fastq" else "_q.fa"))
    open(rf, "w", newline="").write(rtxt); open(qf, "w", newline="").write(qtxt)
    outs = {}
    for tag, exe in (("hip", HIP), ("ref", U.REF_BIN)):
        o = os.path.join(td, tag + ".paf")
        if os.path.exists(o): os.remove(o)
        try:
            p = subprocess.run([exe, "-r", rf, "-q", qf, "-t", "4", "-s", "5000", "--pi", "85", "-o", o], capture_output=True, text=True, timeout=120)
            outs[tag] = (p.returncode, open(o, "rb").read() if os.path.exists(o) else b"<no file>", p.stderr[-300:])
        except subprocess.TimeoutExpired:
            outs[tag] = ("timeout", b"", "")
    same_rc = (outs["hip"][0] == 0) == (outs["ref"][0] == 0)
    ok = same_rc and (outs["hip"][0] != 0 or outs["hip"][1] == outs["ref"][1])
    bad += 0 if ok else 1
    print("ok  " if ok else "DIFF", name, "rc hip/ref", outs["hip"][0], outs["ref"][0], "lines", outs["ref"][1].count(b"\n"), flush=True)
    if not ok:
        print("   hip:", outs["hip"][1][:200], "|", outs["hip"][2][-200:].replace("\n", " / "))
        print("   ref:", outs["ref"][1][:200], "|", outs["ref"][2][-200:].replace("\n", " / "), flush=True)
print("edge inputs done: %d differences" % bad)
