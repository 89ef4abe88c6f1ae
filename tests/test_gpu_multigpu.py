"""The multi-GPU product path on whatever GPUs the box has: reads sharded in contiguous blocks over one
mm_ctx per GPU, index replicated GPU to GPU, one all-gatherv of the candidate mappings (mm_comm.hip), CPU filters afterwards.

A 1-GPU box still runs all of it: two contexts on one device form a local group (the exchange is then device copies instead of
RCCL broadcasts, everything else is the same code), and a world-size-1 RCCL communicator exercises the RCCL leg itself.
With >= 2 GPUs the same tests also run over distinct devices (RCCL over xGMI)."""
import os
import subprocess

import numpy as np
import pytest

import mmutil as U
from golden import cases as CS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_BIN = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
PAF_DIR = os.path.join(ROOT, "tests", "golden", "paf")
CASES = {c[0]: c for c in CS.paf_cases()}


def _ngpu():
    import torch
    return torch.cuda.device_count()


def _device_lists():
    out = [[0, 0], [0, 0, 0]]
    if _ngpu() >= 2:
        out += [[0, 1], list(range(min(_ngpu(), 8)))]
    return out


def _setup(devs, contigs, k=19, L=5000, s=130, pi=0.85):
    from mashmap_amd import capi
    ctxs = [capi.Context(k=k, segLength=L, sketchSize=s, device=d) for d in devs]
    ctxs[0].index_build([a for _, a in contigs], kmerPct=0.001)
    for c in ctxs[1:]:
        c.index_replicate_from(ctxs[0])
    for c in ctxs:
        c.set_tables_default(pi)
    return ctxs


@pytest.mark.parametrize("devs", _device_lists(), ids=lambda d: "dev" + "_".join(map(str, d)))
def test_sharded_batch_gathers_to_the_single_gpu_records(devs):
    """the gathered candidate mappings of a sharded batch == the records one context produces for the whole batch, byte for byte"""
    from mashmap_amd import capi, shard
    contigs = [("c0", U.random_dna(1, 400000)), ("c1", U.random_dna(2, 300000))]
    reads = [a for _, a, _ in U.sample_reads([c for _, c in contigs], 3, 61, 10000, 0.10)]
    reads += [a for _, a, _ in U.sample_reads([c for _, c in contigs], 4, 9, 23456, 0.05)] + [U.random_dna(5, 300), U.random_dna(6, 12)]
    ctxs = _setup(devs, contigs)
    ctxs[0].reads_upload(reads, seqCounterBase=100)
    ctxs[0].map()
    single = ctxs[0].mappings()
    assert len(single) > 100
    capi.comm_init_local(ctxs)
    blocks = [shard.read_block_by_bases([len(r) for r in reads], r, len(ctxs)) for r in range(len(ctxs))]
    for c, (a, b) in zip(ctxs, blocks):
        c.reads_upload(reads[a:b], seqCounterBase=100 + a)
        c.map()
    capi.allgatherv_mappings_local(ctxs)
    for c in ctxs:
        got, counts = c.gathered(len(ctxs))
        assert counts.sum() == len(single) and got.tobytes() == single.tobytes()
    for c in ctxs:
        c.close()


def test_rccl_communicator_world_of_one():
    """the RCCL leg on a single GPU: unique id, ncclCommInitRank, count all-gather + grouped broadcasts"""
    from mashmap_amd import capi
    contigs = [("c0", U.random_dna(11, 300000))]
    reads = [a for _, a, _ in U.sample_reads([c for _, c in contigs], 12, 20, 10000, 0.08)]
    (ctx,) = _setup([0], contigs)
    ctx.comm_init_rank(capi.comm_unique_id(), 0, 1)
    ctx.reads_upload(reads)
    ctx.map()
    ctx.allgatherv_mappings()
    got, counts = ctx.gathered(1)
    mine = ctx.mappings()
    assert len(mine) >= 20 and list(counts) == [len(mine)] and got.tobytes() == mine.tobytes()
    ctx.reads_upload([U.random_dna(13, 9000)])                       # a batch without a single mapping
    ctx.map()
    ctx.allgatherv_mappings()
    got, counts = ctx.gathered(1)
    assert len(got) == 0 and list(counts) == [0]
    ctx.close()


def test_overlapped_exchange_world_of_one():
    """mm_allgatherv_mappings_begin / _end: the exchange of batch A runs while batch B is uploaded and mapped; afterwards the gathered
    records are A's and the resident ones are B's"""
    from mashmap_amd import capi
    contigs = [("c0", U.random_dna(21, 400000))]
    A = [a for _, a, _ in U.sample_reads([c for _, c in contigs], 22, 40, 10000, 0.08)]
    B = [a for _, a, _ in U.sample_reads([c for _, c in contigs], 23, 25, 7000, 0.05)]
    (ctx,) = _setup([0], contigs)
    ctx.comm_init_rank(capi.comm_unique_id(), 0, 1)
    with pytest.raises(capi.MashmapError):
        ctx.allgatherv_mappings_end()                                 # nothing in flight
    for rnd in range(3):
        ctx.reads_upload(A); ctx.map()
        mineA = ctx.mappings()
        ctx.allgatherv_mappings_begin()
        with pytest.raises(capi.MashmapError):
            ctx.allgatherv_mappings_begin()                           # one exchange in flight per context
        with pytest.raises(capi.MashmapError):
            ctx.allgatherv_mappings()
        ctx.reads_upload(B); ctx.map()
        mineB = ctx.mappings()
        ctx.allgatherv_mappings_end()
        got, counts = ctx.gathered(1)
        assert len(mineA) >= 40 and len(mineB) >= 25 and mineA.tobytes() != mineB.tobytes()
        assert list(counts) == [len(mineA)] and got.tobytes() == mineA.tobytes()
        assert ctx.mappings().tobytes() == mineB.tobytes()
    ctx.allgatherv_mappings_begin()                                   # a context destroyed with an exchange in flight joins it
    ctx.close()


def _run(td, name, refrec, qrec, extra, tag, devices, threads="4", env_extra=None):
    rf = os.path.join(td, name + ".ref.fa")
    if not os.path.exists(rf):
        U.write_fasta(rf, refrec)
    out = os.path.join(td, "%s.%s.paf" % (name, tag))
    args = [HIP_BIN, "-r", rf, "-o", out, "-t", threads] + extra
    if qrec is not None:
        qf = os.path.join(td, name + ".q.fa")
        if not os.path.exists(qf):
            U.write_fasta(qf, qrec)
        args += ["-q", qf]
    env = dict(os.environ)
    env["MASHMAP_HIP_DEVICES"] = devices
    env.update(env_extra or {})
    p = subprocess.run(args, capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    return open(out, "rb").read()


@pytest.mark.parametrize("name", ["default", "asm_one2one", "allvsall_Y", "dense_pi80"])
def test_paf_of_a_sharded_run_is_the_single_gpu_paf(name, tmp_path):
    """MASHMAP_HIP_DEVICES: the command line sharding every batch over several contexts writes the golden PAF (== the reference's)
    byte for byte, the one-to-one filter over the gathered mappings included"""
    _, refrec, qrec, extra = CASES[name]
    exp = open(os.path.join(PAF_DIR, name + ".paf"), "rb").read()
    lists = ["0,0", "0,0,0"] + (["0,1"] if _ngpu() >= 2 else [])
    for devs in lists:
        got = _run(str(tmp_path), name, refrec, qrec, extra, "d" + devs.replace(",", ""), devs)
        assert got == exp, "MASHMAP_HIP_DEVICES=%s differs from the golden PAF" % devs
    # small batches: several exchanges per run, some blocks empty (the batch size is per context: 3 x 0.01 Mbp)
    got = _run(str(tmp_path), name, refrec, qrec, extra, "small", "0,0,0", env_extra={"MASHMAP_HIP_BATCH_MBP": "0.01"})
    assert got == exp
    # the device-side all-gatherv in front of a single download (the default hands every context's block to the host directly)
    for devs, mbp in (("0,0", "512"), ("0,0,0", "0.01")):
        got = _run(str(tmp_path), name, refrec, qrec, extra, "gather" + devs.replace(",", ""), devs, env_extra={"MASHMAP_HIP_EXCHANGE": "allgather", "MASHMAP_HIP_BATCH_MBP": mbp})
        assert got == exp, "MASHMAP_HIP_EXCHANGE=allgather with MASHMAP_HIP_DEVICES=%s differs from the golden PAF" % devs
