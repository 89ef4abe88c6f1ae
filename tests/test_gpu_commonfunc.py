"""skch::CommonFunc::sketchSequence / addMinmers (mashmap_amd/host/skch_commonfunc.hpp: the reference's template signatures,
commonFunc.hpp:183 and :302, over the C ABI) against the oracle: a C++ caller is compiled here and run on the GPU."""
import os
import subprocess

import numpy as np
import pytest

import mmutil as U
from test_dropin_compile import _build_commonfunc_check

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("k,w,s", [(19, 1000, 40), (16, 500, 25)])
def test_commonfunc_seams_match_the_oracle(tmp_path, k, w, s):
    orc = U.Oracle()
    seqs = [U.random_dna(11, 300), U.random_dna(12, 5000), U.lowercase_some(U.random_dna(13, 7000), 3), U.with_n_runs(U.random_dna(14, 9000), 5, 4, 300),
            U.tandem_repeat(15, 4000, 37), U.random_dna(16, w - 1), U.random_dna(17, k - 1)]
    seqs[1][3] = ord("N")                                     # an N among the first k-1 bases: addMinmers has no initial-N scan (commonFunc.hpp:334)
    path = tmp_path / "seqs.txt"
    path.write_bytes(b"".join(a.tobytes() + b"\n" for a in seqs))
    exe = _build_commonfunc_check(str(tmp_path / "cfc"), False)
    p = subprocess.run([exe, str(k), str(w), str(s), str(path)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-400:]
    got = {"S": {}, "M": {}}
    for line in p.stdout.splitlines():
        f = line.split()
        if f[0] == "N":
            assert f[2] != "normalisation", line
            continue
        got[f[0]].setdefault(int(f[1]), []).append(tuple(int(x) for x in f[2:]))
    for i, a in enumerate(seqs):
        exp_s = orc.sketch_sequence(a, k, s, 100 + i)
        assert got["S"].get(i, []) == exp_s, ("sketchSequence", i, got["S"].get(i, [])[:3], exp_s[:3])
        m = orc.add_minmers(a, k, w, s, 7 + i)
        exp_m = [(int(x["hash"]), int(x["wpos"]), int(x["wpos_end"]), int(x["seqId"]), int(x["strand"])) for x in m]
        assert got["M"].get(i, []) == exp_m, ("addMinmers", i, len(got["M"].get(i, [])), len(exp_m))
