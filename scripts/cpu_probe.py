import os, sys, time, subprocess, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import bench as B
dev = torch.device('cuda', 0)
contigs = B.make_reference(torch, dev, B.REF_CONTIGS, B.REF_CONTIG_LEN)
ref_np = [c.cpu().numpy() for c in contigs]
n = 10000
reads = B.make_reads(torch, dev, contigs, n, B.READ_LEN, B.ERR, seed=1000).cpu().numpy().reshape(n, B.READ_LEN)
td = tempfile.mkdtemp()
rp, qp = td + '/r.fa', td + '/q.fa'
B.write_fasta(rp, ['chr%d' % i for i in range(len(ref_np))], ref_np)
B.write_fasta(qp, ['read%d' % i for i in range(n)], list(reads))
open(qp + '.fai', 'w').write(''.join('read%d\t%d\t0\t100\t101\n' % (i, B.READ_LEN) for i in range(n)))
print('cores', os.cpu_count())
for t in (32,):
    t0 = time.time()
    p = subprocess.run(['oracle/_ref/mashmap_ref', '-r', rp, '-q', qp, '-o', td + '/o.paf', '-t', str(t), '-J', '130'], capture_output=True, text=True)
    tm = [l for l in p.stderr.splitlines() if 'time spent' in l]
    print(t, round(time.time() - t0, 1), tm)
# our CLI end to end on the same files
for t in (32,):
    t0 = time.time()
    os.environ['MASHMAP_HIP_TIMING']='1'
    p = subprocess.run(['mashmap_amd/lib/mashmap_hip', '-r', rp, '-q', qp, '-o', td + '/h.paf', '-t', str(t), '-J', '130'], capture_output=True, text=True)
    print('hip', t, round(time.time() - t0, 1), [l for l in p.stderr.splitlines() if 'time spent' in l or 'timing' in l])
print('paf identical:', open(td + '/o.paf','rb').read() == open(td + '/h.paf','rb').read(), sum(1 for _ in open(td+'/h.paf')))
