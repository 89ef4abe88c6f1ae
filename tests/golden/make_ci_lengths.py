#!/usr/bin/env python3
"""tests/golden/scerevisiae8_lengths.tsv: names and lengths of the 136 sequences of the reference's CI input (data/scerevisiae8.fa.gz --
the FASTA itself is not in the reference tree, its .fai index is): `name<TAB>length` per line, from the first two columns of
/root/reference/data/scerevisiae8.fa.gz.fai.  Run in the build container:  python tests/golden/make_ci_lengths.py"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
with open("/root/reference/data/scerevisiae8.fa.gz.fai") as f, open(os.path.join(HERE, "scerevisiae8_lengths.tsv"), "w") as o:
    for line in f:
        name, length = line.split("\t")[:2]
        o.write("%s\t%d\n" % (name, int(length)))
