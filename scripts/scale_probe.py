"""configs[2]-style sanity at a larger scale than the unit tests: assembly vs reference (100 Mbp contigs), pi = 95,
segLength 10000, one-to-one filter, through the mashmap_hip CLI and the stock binary; PAF must be byte-identical."""
import os, sys, time, subprocess, tempfile
sys.path.insert(0, '.')
import numpy as np, torch
import bench as B
dev = torch.device('cuda', 0)
NC, CL = int(os.environ.get("NC", 3)), int(os.environ.get("CL", 100_000_000))
g = torch.Generator(device=dev); g.manual_seed(7)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
comp = torch.zeros(256, dtype=torch.uint8, device=dev)
for a, b in zip(b"ACGT", b"TGCA"): comp[a] = b
td = tempfile.mkdtemp()
ref, qry = [], []
for i in range(NC):
    c = lut[torch.randint(0, 4, (CL,), generator=g, device=dev)]
    if i == 1: c[CL // 2:CL // 2 + 200000] = ord('N')                      # an assembly gap
    m = torch.rand(CL, generator=g, device=dev) < 0.01
    q = torch.where(m, lut[torch.randint(0, 4, (CL,), generator=g, device=dev)], c)
    if i == 2: q = comp[q.flip(0).long()]
    if i == 0: q = torch.cat([q[30_000_000:60_000_000], q[:30_000_000], q[60_000_000:]])   # a translocation
    ref.append(c.cpu().numpy()); qry.append(q.cpu().numpy())
rp, qp = td + '/ref.fa', td + '/qry.fa'
B.write_fasta(rp, ['chr%d' % i for i in range(NC)], ref)
B.write_fasta(qp, ['ctg%d' % i for i in range(NC)], qry)
args = ['--pi', '95', '-s', '10000', '-f', 'one-to-one', '-J', '40', '-t', os.environ.get('THREADS', '32')]
print('scale probe: %d contigs x %d bp, mashmap %s' % (NC, CL, ' '.join(args)))
out = {}
for name, exe in (('hip', 'mashmap_amd/lib/mashmap_hip'), ('ref', 'oracle/_ref/mashmap_ref')):
    if not os.path.exists(exe): continue
    t0 = time.time()
    p = subprocess.run([exe, '-r', rp, '-q', qp, '-o', td + '/%s.paf' % name] + args, capture_output=True, text=True)
    print(name, 'rc', p.returncode, 'wall %.1f s' % (time.time() - t0), [l.split('] ')[-1] for l in p.stderr.splitlines() if 'time spent' in l])
    if p.returncode: print(p.stderr[-1500:])
    out[name] = open(td + '/%s.paf' % name, 'rb').read()
print('lines', {k: v.count(b'\n') for k, v in out.items()})
if len(out) == 2:
    print('PAF identical:', out['hip'] == out['ref'])
    if out['hip'] != out['ref']:
        a, b = out['hip'].decode().splitlines(), out['ref'].decode().splitlines()
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y: print(i, x, '\n ', y); break
print(out.get('hip', b'').decode()[:600])
