cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r05c; mkdir -p $OUT
cat /sys/fs/cgroup/cpu.stat > $OUT/cgroup_before.txt 2>&1
python scripts/e2e_fasta_paf.py > $OUT/a_default.json 2> $OUT/a_default.err
cat /sys/fs/cgroup/cpu.stat > $OUT/cgroup_after_a.txt 2>&1
run() { name=$1; shift; env "$@" python scripts/e2e_fasta_paf.py --reuse > $OUT/$name.json 2> $OUT/$name.err; }
run b_r10_p6 MASHMAP_HIP_READER_THREADS=10 MASHMAP_HIP_POST_THREADS=6
run c_r12_p4 MASHMAP_HIP_READER_THREADS=12 MASHMAP_HIP_POST_THREADS=4
run d_r8_p8 MASHMAP_HIP_READER_THREADS=8 MASHMAP_HIP_POST_THREADS=8
run e_r14_p8 MASHMAP_HIP_READER_THREADS=14 MASHMAP_HIP_POST_THREADS=8
run f_r12_p8 MASHMAP_HIP_READER_THREADS=12 MASHMAP_HIP_POST_THREADS=8
run g_r16_p16 MASHMAP_HIP_READER_THREADS=16 MASHMAP_HIP_POST_THREADS=16
run h_default MASHMAP_HIP_STALL_TRACE=1
cat /sys/fs/cgroup/cpu.stat > $OUT/cgroup_after_h.txt 2>&1
for f in a_default b_r10_p6 c_r12_p4 d_r8_p8 e_r14_p8 f_r12_p8 g_r16_p16 h_default; do echo $f; cut -c330-640 $OUT/$f.json; grep -c "stall\]" $OUT/$f.err; done
grep -h "thrott" $OUT/cgroup_*.txt
