cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r04j; mkdir -p $OUT
MASHMAP_HIP_READER_THREADS=8 python scripts/e2e_fasta_paf.py > $OUT/a_default.json 2> $OUT/a_default.err
MASHMAP_HIP_READER_THREADS=8 MASHMAP_HIP_BIG_BUFFERS=1 python scripts/e2e_fasta_paf.py --reuse > $OUT/b_bigbuf.json 2> $OUT/b_bigbuf.err
MASHMAP_HIP_READER_THREADS=8 MASHMAP_HIP_NO_PREFAULT=1 python scripts/e2e_fasta_paf.py --reuse > $OUT/c_noprefault.json 2> $OUT/c_noprefault.err
MASHMAP_HIP_READER_THREADS=8 python scripts/e2e_fasta_paf.py --reuse > $OUT/d_default_again.json 2> $OUT/d_default_again.err
MASHMAP_HIP_READER_THREADS=8 MASHMAP_HIP_BIG_BUFFERS=1 python scripts/e2e_fasta_paf.py --reuse > $OUT/e_bigbuf_again.json 2> $OUT/e_bigbuf_again.err
MASHMAP_HIP_READER_THREADS=16 python scripts/e2e_fasta_paf.py --reuse > $OUT/f_rt16.json 2> $OUT/f_rt16.err
for f in a_default b_bigbuf c_noprefault d_default_again e_bigbuf_again f_rt16; do echo $f; cut -c330-640 $OUT/$f.json; done
numactl -H 2>/dev/null | head -5; lscpu | grep -i "numa\|socket\|model name" | head
