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
               INDELMINER_RENDEZVOUS=os.environ.get("IM_TEST_RENDEZVOUS") or os.path.join(cwd, "rdv_%d" % port),
               INDELMINER_RUN_TOKEN="test-%d" % port)          # the ranks are started by different processes here
    r = subprocess.run([binary] + flags + [ref, "sample=" + bam], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    # every rank must have finished cleanly (IM_TEST_RC0: rank 0 is expected to end with that status); gloo carries the verdict to rank 0
    good = int(os.environ.get("IM_TEST_RC0", "0")) if rank == 0 else 0
    ok = torch.tensor([1 if r.returncode == good else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put((int(ok[0]), r.stdout, r.stderr[-3000:]))
    elif r.returncode != 0:
        sys.stderr.write(r.stderr.decode()[-3000:])
    dist.destroy_process_group()


def _run_world(world, flags, cwd, ref, bam, with_stderr=False):
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
    return (out, err) if with_stderr else out


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
    for world in (2, 5):                # 5 = an idle rank (3 ranks: test_three_ranks_two_contigs and the pieces tests)
        assert _run_world(world, [], str(tmp_path), "ref.fa", "aln.bam") == want, world


@pytest.mark.parametrize("flags", [["-i", "cfg.txt"], []])
def test_ranks_with_many_waiting_mates_that_are_not_stale(tmp_path, flags):
    """more than forty waiting first mates of contig 0 fail the |isize| > range[1] test (src/indelminer.c:519) and lie in front of
    the few that pass it and pin every later marker: the exchange carries every pair-table record and each rank replays the table
    with the final insert lengths -- with a config file and with the table estimated in the same exchange"""
    d = th._many_waiting_dir(tmp_path)
    want = th._run(th._build_shim(), flags, d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"\n") > 100
    assert th._run(th._build_shim(), flags, d, ref="ref.fa", bam="aln.bam") == want        # the walker pool of one process
    for world in (2, 3):
        assert _run_world(world, flags, d, "ref.fa", "aln.bam") == want, world


def test_ranks_with_names_shared_between_contigs(tmp_path):
    """every rank sees in the exchanged logs that an entry of contig 0 meets a record of contig 2: rank 0 hands the run to one
    record-at-a-time process (the reference's one pair table), the other ranks leave"""
    d = th._shared_names_dir(tmp_path)
    want = th._run(th._build_shim(), [], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert _run_world(2, [], d, "ref.fa", "aln.bam") == want
    assert _run_world(3, [], d, "ref.fa", "aln.bam") == want


def test_ranks_ignore_what_an_earlier_run_left_in_the_rendezvous_directory(tmp_path_factory, monkeypatch):
    """a re-used rendezvous directory with another run's id file, part files and flags: the id is not this run's (token), rank 0
    empties the directory before it publishes its own, and no stale part reaches the output"""
    d = th._synth_dir(tmp_path_factory, "synth_2ctg_composite")
    rdv = os.path.join(d, "rdv_reused")
    os.makedirs(rdv, exist_ok=True)
    open(os.path.join(rdv, "rccl_id"), "wb").write(b"/tmp/im_shim_comm_of_another_run".ljust(128, b"\0") + b"another-run".ljust(96, b"\0"))
    open(os.path.join(rdv, "part.1"), "w").write("ctg1\t1\t.\tA\tC\tSTALE\n")
    open(os.path.join(rdv, "done.1"), "w").write("-1\n")
    monkeypatch.setenv("IM_TEST_RENDEZVOUS", rdv)
    assert _run_world(2, ["-i", "cfg.txt"], d, "ref.fa", "aln.bam") == th._golden("synth_2ctg_composite")


def test_log_that_does_not_fit_the_first_exchange(tmp_path):
    """a rank whose pair-table log is longer than the exchange buffer says so in its header and the exchange is repeated once
    with the size that fits"""
    d = th._stale_dir(tmp_path)
    want = th._run(th._build_shim(), [], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    os.environ["INDELMINER_MG_LOG_BYTES"] = "4800"
    try:
        assert _run_world(2, [], d, "ref.fa", "aln.bam") == want
    finally:
        del os.environ["INDELMINER_MG_LOG_BYTES"]


def test_ranks_hand_a_run_the_reference_aborts_to_one_process(tmp_path):
    """a record the reference dies on (an N op in a proper pair, new_readseg_bam) in the middle of contig 2 of 4: the rank that owns
    the contig reports it, rank 0 prints the parts in front of that contig and starts the record-at-a-time child, which prints
    the flushes in front of the record and dies with the reference's message and status -- the single run's bytes and status"""
    import numpy as np
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=54, ref_len=220_000, coverage=30, n_contigs=4, big_every=3)
    proper = ((rd.flag & 0x2) != 0) & (rd.tid == 2) & (rd.pos > 150_000) & (rd.ncig == 1)
    i = int(np.nonzero(proper)[0][5])
    rd.cig_op[i, 0] = 3
    contigs = [("ctg%d" % k, len(r)) for k, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    one = subprocess.run([th._build_shim(), "ref.fa", "s=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=dict(os.environ, INDELMINER_PIPELINE="host"))
    assert one.returncode == 1 and b"new_readseg_bam" in one.stderr and one.stdout.count(b"\n") > 60
    os.environ["IM_TEST_RC0"] = "1"
    try:
        for world in (2, 3):
            assert _run_world(world, [], str(tmp_path), "ref.fa", "s=aln.bam".split("=")[1]) == one.stdout, world
    finally:
        del os.environ["IM_TEST_RC0"]


def test_ranks_hand_over_an_aborted_run_whose_contigs_lie_in_pieces_over_the_ranks(tmp_path):
    """the same record with the contigs cut into pieces that every rank walks (walked groups travel to the contig's owner, the
    depth arrays are summed over the ranks): the rank that meets the record must not leave the others waiting in a collective --
    every rank learns of the aborted walk in the exchange, stops in front of that contig, joins the sum and reports; rank 0
    prints what lies in front and starts the record-at-a-time child.  The single run's bytes and status, within seconds."""
    import time
    import numpy as np
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=54, ref_len=220_000, coverage=30, n_contigs=4, big_every=3)
    proper = ((rd.flag & 0x2) != 0) & (rd.tid == 2) & (rd.pos > 150_000) & (rd.ncig == 1)
    i = int(np.nonzero(proper)[0][5])
    rd.cig_op[i, 0] = 3
    contigs = [("ctg%d" % k, len(r)) for k, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    one = subprocess.run([th._build_shim(), "ref.fa", "s=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=dict(os.environ, INDELMINER_PIPELINE="host"))
    assert one.returncode == 1 and b"new_readseg_bam" in one.stderr and one.stdout.count(b"\n") > 60
    env = {"IM_TEST_RC0": "1", "INDELMINER_MG_FORCE_SPLIT": "1", "INDELMINER_PIECE_BYTES": "150000", "INDELMINER_MG_TIMEOUT": "60"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for world in (2, 3):
            t = time.time()
            assert _run_world(world, [], str(tmp_path), "ref.fa", "aln.bam") == one.stdout, world
            assert time.time() - t < 50, "a rank sat out its watchdog"
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_one_rank_ships_every_group_to_itself(tmp_path_factory):
    """INDELMINER_MG_SELF_SHIP=1: every claim takes the road of a claim walked for another rank's contig -- the group's host
    part serialised, host part and device arrays through the send / receive group (here to the rank itself), unpacked, staged
    from the arrived arrays: the single run's bytes (the same test runs on the GPU box with the real RCCL group)"""
    d = th._synth_dir(tmp_path_factory, "synth_2ctg_composite")
    env = {"INDELMINER_FORCE_MGPU": "1", "INDELMINER_MG_SELF_SHIP": "1", "INDELMINER_PIECE_BYTES": "60000",
           "INDELMINER_RENDEZVOUS": os.path.join(d, "rdv_self"), "INDELMINER_RUN_TOKEN": "self"}
    got = th._run(th._build_shim(), ["-i", "cfg.txt"], d, ref="ref.fa", bam="aln.bam", env=env)
    assert got == th._golden("synth_2ctg_composite")


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("world", [2, 3])
def test_one_contig_over_several_ranks(tmp_path_factory, world):
    """ONE contig, several ranks: its pieces are walked (read, inflated, triaged) by all of them, the walked groups travel to the
    contig's owner, which serves the pair table, places the markers, runs the stage and replays -- with the depth arrays of the
    ranks summed in one all-reduce first.  The bytes of the single-process run, with and without a config file."""
    d = th._synth_dir(tmp_path_factory, "synth_1mb_30x")
    runs = (((["-i", "cfg.txt"], "synth_1mb_30x", "400000"), ([], "synth_1mb_30x_noconfig", "90000")) if world == 2 else
            ((["-i", "cfg.txt"], "synth_1mb_30x", "90000"), ([], "synth_1mb_30x_noconfig", "400000")))
    for flags, golden, pb in runs:
        got = _with_env({"INDELMINER_PIECE_BYTES": pb}, lambda: _run_world(world, flags, d, "ref.fa", "aln.bam"))
        assert got == th._golden(golden), (flags, pb)


def test_long_read_library_over_two_ranks(tmp_path):
    """a 2 x 1100 library and a 2 x 300 library with -g 2, ONE contig in pieces over two ranks: the owner realigns groups another rank
    walked, so it must hear of their longest read from the package (the realign launches for long reads / the general pass follow
    im_expect_read_length).  The bytes of the record-at-a-time run."""
    for sub, read_len, flags in (("a", 1100, ["-i", "cfg.txt"]), ("b", 300, ["-i", "cfg.txt", "-g", "2"])):
        (tmp_path / sub).mkdir()
        d = th._long_read_dir(tmp_path / sub, read_len=read_len, ref_len=120_000, coverage=14, seed=70 + read_len)
        want = th._run(th._build_shim(), flags, d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
        assert want.count(b"SPLIT_READ") > 10
        got = _with_env({"INDELMINER_PIECE_BYTES": "150000", "INDELMINER_MG_FORCE_SPLIT": "1"}, lambda: _run_world(2, flags, d, "ref.fa", "aln.bam"))
        assert got == want, sub


def test_contigs_in_pieces_over_ranks_with_markers_pinned_low(tmp_path):
    """four contigs, first mates that wait for ever in two of them (every later marker pinned, also in later contigs), pieces of
    every contig on every rank: counter prefixes per piece, marker floors per contig, frozen evidence waiting for its contig's end"""
    d = th._stale_dir(tmp_path)
    want = th._run(th._build_shim(), [], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    for world in (3,):
        got = _with_env({"INDELMINER_PIECE_BYTES": "120000"}, lambda: _run_world(world, [], d, "ref.fa", "aln.bam"))
        assert got == want, world
    det = th._run(th._build_shim(), ["-o", "detailed"], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert _with_env({"INDELMINER_PIECE_BYTES": "200000"}, lambda: _run_world(2, ["-o", "detailed"], d, "ref.fa", "aln.bam")) == det


def test_ranks_coverage_table(tmp_path):
    """no config file: every rank's span sums and covered segments ride in the one all-gather; rank 0 prints the reference's table"""
    d, want = th._coverage_tables(th._build_shim(), tmp_path, ())
    for world, env in ((2, {}), (3, {"INDELMINER_PIECE_BYTES": "60000"})):
        out, err = _with_env(env, lambda: _run_world(world, [], d, "ref.fa", "aln.bam", with_stderr=True))
        assert th._coverage_table(err) == want, (world, err[-800:])
