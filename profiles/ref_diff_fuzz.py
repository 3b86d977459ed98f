#!/usr/bin/env python3
"""CPU soak in the build container: the whole program against the COMPILED REFERENCE (oracle/_ref/indelminer) on small synthetic
inputs whose records are then made odd -- flag bits, strands, insert sizes, mate contigs, =/X ops, N / H / P ops, clips inside the
CIGAR, base codes, MQ / RG tags of right and wrong types, mapping qualities.  The product runs through the CPU shim (the device
entry points answered by the oracle) in pipeline mode and record-at-a-time; exit code and stdout must be the reference's.
    python profiles/ref_diff_fuzz.py [first_seed] [n_inputs] [workers]"""
import hashlib, os, random, shutil, subprocess, sys, tempfile
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "indelminer")


def one(seed):
    from indelminer_amd import bamwrite, synth
    from tests.test_host_driver import _build_shim
    from tests.golden.odd_inputs import make_input, mutate
    d = tempfile.mkdtemp(prefix="rdf%d_" % seed, dir="/tmp")
    try:
        cmd, info = make_input(seed, d)
        rng, refs, contigs, nct, args = info["rng"], info["refs"], info["contigs"], info["nct"], info["args"]
        outs = []
        shim = os.environ.get("RDF_SHIM") or _build_shim()      # RDF_SHIM: a sanitizer build of the same sources
        for b, env in ((REF, {}), (shim, {}), (shim, {"INDELMINER_PIPELINE": "host"}), (shim, {"INDELMINER_WALKERS": "1", "INDELMINER_REPLAYERS": "1"} if seed & 1 else
                             {"INDELMINER_CLAIM_BASES": "1", "INDELMINER_WALKERS": "3", "INDELMINER_REPLAYERS": "2"})):     # every contig a group of its own
            try:
                r = subprocess.run([b] + cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=600)
                err = r.stderr.decode(errors="replace").strip().splitlines()
                rc = "sanitizer" if b"Sanitizer" in r.stderr else r.returncode
                outs.append((rc, hashlib.md5(r.stdout).hexdigest(), len(r.stdout), err[-1][:100] if err else ""))
            except subprocess.TimeoutExpired:
                outs.append(("timeout", "", 0, ""))
        signalled = isinstance(outs[0][0], int) and outs[0][0] < 0
        if signalled:
            # the reference died of a signal (strlen(NULL) on a read-group tag of another type, an assert inside samtools): the product
            # exits with status 1 and a message there
            outs[0] = (1,) + outs[0][1:]
        same = all((o[0], o[1]) == (outs[0][0], outs[0][1]) for o in outs[1:])
        # A run the reference aborts: same exit status, and the same stdout in front of the abort (the header, the variants of the
        # flushes in front of the fatal record) in every mode -- the device pipeline hands such a run over to a record-at-a-time
        # child.  A reference that died of a signal lost its buffered stdout.
        if not same and outs[0][0] != 0:
            same = all(o[0] == outs[0][0] for o in outs[1:]) and (signalled or all(o[1] == outs[0][1] for o in outs[1:]))
        if same and outs[0][0] == 0 and "-c" not in args and rng.random() < 0.4:
            # the multi-GPU mode of the CLI (one process per rank, contigs tid % world, one exchange of summaries), ranks as plain
            # processes on the shim: rank 0's stdout must be the single-process VCF
            world = rng.choice([2, 3])
            rdv = os.path.join(d, "rdv")
            procs = []
            for rank in range(world):
                env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(20000 + seed % 20000),
                           INDELMINER_RENDEZVOUS=rdv)
                procs.append(subprocess.Popen([shim] + cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env))
            res = [p.communicate(timeout=600) for p in procs]
            rcs = [p.returncode for p in procs]
            mg = (max(rcs) if min(rcs) >= 0 else min(rcs), hashlib.md5(res[0][0]).hexdigest(), len(res[0][0]), "world %d" % world)
            if (mg[0], mg[1]) != (outs[0][0], outs[0][1]):
                same = False
                outs = outs + [("multi-rank",) + mg + (res[0][1].decode(errors="replace").strip().splitlines()[-1:],)]
        if same and outs[0][0] == 0 and "-o" not in args and rng.random() < 0.5:
            # annotate mode (README.md:116 of the reference): the calls of this sample tagged against a second one (other reads of
            # the same genome, made odd the same way)
            r0 = subprocess.run([REF] + cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            open(os.path.join(d, "calls.vcf"), "wb").write(r0.stdout)
            refs2, rd2 = synth.simulate(seed=seed, ref_len=len(refs[0]), coverage=rng.choice([6, 12]), n_contigs=nct, read_seed=seed + 31,
                                        big_every=0, indel_spacing=rng.choice([700, 2000]), somatic_spacing=0)
            mutate(rng, rd2, rng.random() < 0.3)        # some second samples with records the reference dies on
            bamwrite.write_bam(os.path.join(d, "other.bam"), contigs, rd2)
            acmd = [a for a in args if a != "-c" and a != "ctg0"] + ["ref.fa", "calls.vcf", "other=other.bam"]
            aouts = []
            for b, env in ((REF, {}), (shim, {}), (shim, {"INDELMINER_PIPELINE": "host"})):
                r = subprocess.run([b] + acmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=600)
                err = r.stderr.decode(errors="replace").strip().splitlines()
                aouts.append((r.returncode if r.returncode >= 0 else 1, hashlib.md5(r.stdout).hexdigest(), len(r.stdout), err[-1][:100] if err else "", r.returncode < 0))
            # completed: same bytes; aborted: same status (a reference that died of a signal -- its asserts -- lost its buffered stdout)
            if not all(o[0] == aouts[0][0] and (o[1] == aouts[0][1] or aouts[0][4]) for o in aouts[1:]):
                same = False
                outs = outs + [("annotate",) + o for o in aouts]
                cmd = cmd + ["|"] + acmd
        if not same:
            keep = "/tmp/rdf_fail_%d" % seed
            shutil.rmtree(keep, ignore_errors=True); shutil.copytree(d, keep)
        return seed, same, info["fatal_ok"], " ".join(cmd), outs
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    assert os.path.exists(REF), "needs the compiled reference (oracle/Makefile, target ref)"
    from tests.test_host_driver import _build_shim
    _build_shim()
    bad = 0; failed_runs = 0
    with ProcessPoolExecutor(workers) as ex:
        for seed, same, fatal_ok, cmd, outs in ex.map(one, range(first, first + n)):
            failed_runs += outs[0][0] != 0
            if not same:
                bad += 1
                print("DIFF seed %d (%s): %s" % (seed, cmd, outs), flush=True)
            elif seed % 10 == 0:
                print("seed %d ok (rc %s, %d bytes) %s" % (seed, outs[0][0], outs[0][2], outs[0][3][:60]), flush=True)
    print("%d inputs, %d differ, %d where the reference exits with an error" % (n, bad, failed_runs))
    sys.exit(1 if bad else 0)
