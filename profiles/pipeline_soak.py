#!/usr/bin/env python3
"""One-off soak on the GPU box: random synthetic inputs through the product's device pipeline (random walker / claim / replay
settings, one-pass or two-pass) against the product's own one-record-at-a-time host path (INDELMINER_PIPELINE=host), bytes compared.
    python profiles/pipeline_soak.py [first_seed] [n]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from indelminer_amd import bamwrite, build, rawrec, synth

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def run(td, flags, env):
    p = subprocess.run([os.environ.get("IM_SOAK_BIN") or build.HOST_BIN] + flags + ["ref.fa", "s=aln.bam"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, **env))
    return p.returncode, p.stdout, p.stderr[-300:]


for seed in range(first, first + n):
    rng = random.Random(seed)
    nc = rng.choice([1, 2, 3, 5, 7])
    lens = [rng.choice([20_000, 60_000, 150_000, 400_000]) for _ in range(nc)]
    read_len = rng.choice([100, 100, 100, 150, 250, 300])                 # 300: the long-read realign kernel
    refs, rd = synth.simulate(seed=seed, ref_lens=lens, coverage=rng.choice([8, 20, 30, 45]), big_every=rng.choice([0, 3, 7]),
                              indel_spacing=rng.choice([1000, 2000]), read_len=read_len,
                              **({} if read_len <= 150 else dict(isize_mean=3 * read_len, isize_min=3 * read_len - 200, isize_max=3 * read_len + 200)))
    if rng.random() < 0.4:          # second mates of some discordant pairs never come
        both = ((rd.flag & 0x4) == 0) & ((rd.flag & 0x8) == 0)
        drop = both & ((rd.flag & 0x2) == 0) & (rd.pos > rd.mpos) & (rd.pair_id % 4 == 0)
        keep = ~drop
        for name, col in list(vars(rd).items()):
            if isinstance(col, np.ndarray) and len(col) == len(keep):
                setattr(rd, name, col[keep])
        rd.n = int(keep.sum())
    multi_rg = rd.n < 70_000 and rng.random() < 0.5      # several read groups (prefix-aliasing names), the slow writer: small inputs only
    if multi_rg:
        rd.rg_names = rng.sample(["lib10", "li", "", "lib1", "x", "lib", "other_library_with_a_long_name"], rng.choice([2, 3, 4]))
        g = (rd.pair_id % len(rd.rg_names)).astype(np.int64)
        g[rd.tid < rng.randrange(nc)] = 0           # the other groups first appear in a later contig
        rd.rg_idx = g
    with tempfile.TemporaryDirectory() as td:
        contigs = [("c%d" % i, len(r)) for i, r in enumerate(refs)]
        bamwrite.write_fasta(td + "/ref.fa", contigs, refs)
        if multi_rg:
            bamwrite.write_bam(td + "/aln.bam", contigs, rd)
        else:
            rawrec.write_bam_fast(td + "/aln.bam", contigs, rd)
        open(td + "/cfg.txt", "w").write("IL generic 300 %d\n" % rd.range_max)
        flags = [] if multi_rg else rng.choice([[], ["-i", "cfg.txt"], ["-e", "1"], ["-i", "cfg.txt", "-q", "0", "-a"], ["-b", "40", "-n", "15"],
                            ["-g", "2"], ["-k", "8"], ["-o", "detailed"], ["-s", "300"], ["-f", "2"], ["-t"], ["-i", "cfg.txt", "-g", "5", "-k", "5"],
                            ["-c", "c0"], ["-i", "cfg.txt", "-c", "c0:%d-%d" % (lens[0] // 5, lens[0] * 3 // 4)]])
        if read_len > 255 and "-g" in flags:
            flags = ["-i", "cfg.txt"]                                       # long reads run at -g 0 only
        rc0, want, err0 = run(td, flags, {"INDELMINER_PIPELINE": "host", "INDELMINER_ESTIMATE_SERIAL": "1"})
        assert rc0 == 0, err0
        if seed % 10 == 0:          # the multi-GPU code path with one rank (RCCL bring-up: ~2 s)
            mg_env = {"INDELMINER_FORCE_MGPU": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "INDELMINER_RENDEZVOUS": td + "/rdv",
                      "INDELMINER_PIECE_BYTES": str(rng.choice([60_000, 400_000]))}
            if seed % 20 == 0:
                mg_env["INDELMINER_MG_FORCE_SPLIT"] = "1"                   # replays behind the sum of the depth arrays (ncclAllReduce)
            rc, got, err = run(td, flags, mg_env) if "-c" not in flags else (0, want, b"")      # region runs are single-process
            if rc != 0 or got != want:
                print("MISMATCH seed %d flags %r multi-GPU path rc %d: %s" % (seed, flags, rc, err.decode(errors="replace")), flush=True)
                sys.exit(1)
        for _ in range(3):
            env = {"INDELMINER_WALKERS": str(rng.choice([1, 2, 4, 6])), "INDELMINER_REPLAYERS": str(rng.choice([1, 3, 6]))}
            if rng.random() < 0.5:
                env["INDELMINER_CLAIM_BASES"] = str(rng.choice([1, 100_000, 500_000]))
            if rng.random() < 0.4:
                env["INDELMINER_ONEPASS"] = "0"                             # the pre-pass layout (one pass is the default)
            if rng.random() < 0.7:
                env["INDELMINER_PIECE_BYTES"] = str(rng.choice([20_000, 60_000, 150_000, 1_000_000]))
            if rng.random() < 0.3:
                env["INDELMINER_FLUSH_MODE"] = rng.choice(["seq", "per-flush", "wide"])
            if rng.random() < 0.15:
                env["INDELMINER_KEEP_QUAL"] = "1"
            if rng.random() < 0.5:
                env["INDELMINER_SPECULATE"] = "1"                           # one pass staged behind the walk on the first claims' table (may fall back)
            rc, got, err = run(td, flags, env)
            if rc != 0 or got != want:
                print("MISMATCH seed %d flags %r env %r rc %d: %s" % (seed, flags, env, rc, err.decode(errors="replace")), flush=True)
                sys.exit(1)
    print("seed %d ok: %d contigs, %d reads, flags %r, %d VCF bytes%s" % (seed, nc, rd.n, flags, len(want), ", read groups %r" % rd.rg_names if multi_rg else ""), flush=True)
