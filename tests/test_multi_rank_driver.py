"""The multi-GPU product path without GPUs: two ranks of the host driver (linked against the CPU shim,
whose collective is a file-based all-gather), contigs sharded tid % 2, the read-counter prefix / unpaired
mates / insert lengths exchanged in one all-gather, one VCF part per contig, rank 0 concatenating -- must
print the bytes of the single-process run (the reference's golden).  torch.distributed (gloo) launches and
joins the two ranks the way torch.distributed.run would on the GPU node."""
import os
import socket
import subprocess
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_host_driver as th  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, binary, flags, cwd, ref, bam, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_PORT=str(port),
               INDELMINER_RENDEZVOUS=os.path.join(cwd, "rdv_%d" % port))
    r = subprocess.run([binary] + flags + [ref, "sample=" + bam], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    # every rank must have finished cleanly; gloo carries the verdict to rank 0
    ok = torch.tensor([1 if r.returncode == 0 else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put((int(ok[0]), r.stdout, r.stderr[-3000:]))
    elif r.returncode != 0:
        sys.stderr.write(r.stderr.decode()[-3000:])
    dist.destroy_process_group()


def _run_world(world, flags, cwd, ref, bam):
    binary = th._build_shim()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, binary, flags, cwd, ref, bam, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, out, err = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
    assert ok == 1, err.decode()
    return out


@pytest.mark.parametrize("flags,golden", [(["-i", "cfg.txt"], "synth_2ctg_composite"), ([], "synth_2ctg_composite_noconfig")])
def test_two_ranks_print_the_single_run(tmp_path_factory, flags, golden):
    d = th._synth_dir(tmp_path_factory, "synth_2ctg_composite")
    got = _run_world(2, flags, d, "ref.fa", "aln.bam")
    want = th._golden(golden)
    assert got == want


def test_two_ranks_number_the_detailed_blocks_across_the_run(tmp_path_factory):
    """-o detailed numbers its blocks across the contigs (src/variant.c print_det_output's static counter): a rank does not know how
    many blocks the contigs in front of its own print, rank 0 fills the numbers in while it puts the parts together"""
    d = th._synth_dir(tmp_path_factory, "synth_2ctg_composite")
    want = th._run(th._build_shim(), ["-i", "cfg.txt", "-o", "detailed"], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"#####") > 20 and b"\x01" not in want
    assert _run_world(2, ["-i", "cfg.txt", "-o", "detailed"], d, "ref.fa", "aln.bam") == want


def test_three_ranks_two_contigs(tmp_path_factory):
    """more ranks than contigs: the idle rank still takes part in the collectives"""
    d = th._synth_dir(tmp_path_factory, "synth_2ctg_composite")
    got = _run_world(3, ["-i", "cfg.txt"], d, "ref.fa", "aln.bam")
    assert got == th._golden("synth_2ctg_composite")


def test_two_ranks_with_stale_pair_table_entries(tmp_path):
    """contigs 0 and 2 (rank 0's) leave first mates waiting for ever; every later flush marker -- in rank 1's contigs 1 and 3
    too -- is pinned by them in the single run: the marker floor and the read-counter prefix both come from the ONE exchange
    of shard summaries.  No config file, so the insert lengths travel in the same exchange."""
    import numpy as np
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=51, ref_len=250_000, coverage=30, n_contigs=4, big_every=3)
    both = ((rd.flag & 0x4) == 0) & ((rd.flag & 0x8) == 0)
    drop = both & ((rd.flag & 0x2) == 0) & (rd.pos > rd.mpos) & (rd.pair_id % 3 == 0) & np.isin(rd.tid, [0, 2])
    keep = ~drop
    for name, col in list(vars(rd).items()):
        if isinstance(col, np.ndarray) and len(col) == len(keep):
            setattr(rd, name, col[keep])
    rd.n = int(keep.sum())
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    want = th._run(th._build_shim(), [], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"\n") > 100
    for world in (2, 3, 4, 5):          # 4 = one contig per rank, 5 = an idle rank
        assert _run_world(world, [], str(tmp_path), "ref.fa", "aln.bam") == want, world
