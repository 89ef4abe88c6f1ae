"""-Y '#' (skip_prefix) at the reference's CI scale, stage log of mashmap_hip: how much of 'time spent mapping the query' is the device stage when
every fragment takes the per-group literal L1 (computeMap.hpp:1147-1163, :776-782)?  Synthetic 8 x 17 yeast-like sequences (tests/test_gpu_ci_yeast.py),
uncompressed FASTA, with and without -Y."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_ci_yeast as T
seqs = T.ci_sequences()
td = tempfile.mkdtemp()
fa = os.path.join(td, "y8.fa")
with open(fa, "wb") as f:
    for n, a in seqs:
        f.write(b">" + n.encode() + b"\n"); f.write(a.tobytes() + b"\n")
for extra in (["-Y", "#"], ["-X"], []):
    p = subprocess.run([os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip"), "-r", fa, "-q", fa, "--pi", "95", "-n", "1", "-t", "32", "-o", os.path.join(td, "o.paf")] + extra,
                       capture_output=True, text=True, env=dict(os.environ, MASHMAP_HIP_TIMING="1", MM_DEBUG="1"))
    print("==", " ".join(extra) or "(no filter flag)", "rc", p.returncode)
    print("\n".join(l[:230] for l in p.stderr.splitlines() if "device stage" in l or "time spent" in l or "sized pass" in l or "lookup+L1" in l or "reader:" in l or "post stage" in l)[-3000:])
