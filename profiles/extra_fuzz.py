#!/usr/bin/env python3
"""One-off soak on the GPU box: the parity fuzz of tests/test_gpu_realign.py on fresh seeds (g = 0: 40 parameter sets x 60 reads
per seed; g > 0: a re-seeded copy of the gapped fuzz).  Prints a line per seed; exits non-zero at the first difference.
    python profiles/extra_fuzz.py [first_seed] [n_seeds]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indelminer_amd import capi
from tests import test_gpu_realign as T
from tests.support import oraclebind as ob

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = capi.Context(0)
for seed in range(first, first + n):
    T.test_hip_matches_oracle_fuzzed_parameters(ctx, seed)
    print("g=0 fuzz seed %d ok" % seed, flush=True)
    T.test_hip_long_reads_fuzzed(ctx, seed)
    print("long-read fuzz seed %d ok" % seed, flush=True)
for seed in range(first, first + n):
    rng = random.Random(seed)
    for trial in range(10):
        kw = dict(klength=rng.choice([4, 6, 6, 8, 12]), numgaps=rng.choice([1, 2, 3, 5, 8, 12, 20, 40, 60]),
                  maxdelsize=rng.choice([300, 1000]), ethreshold=rng.choice([5, 10]))
        clen = rng.choice([900, 4000, 20000])
        contig = "".join(rng.choice("ACGT") for _ in range(clen))
        cases = []
        for _ in range(40):
            L = min(rng.choice([36, 76, 100, 150, 250]), clen - 10)
            Rm = rng.choice([200, 705])
            anchor = rng.randint(0, clen - 1)
            p = max(0, min(clen - L, anchor + rng.randint(-Rm, Rm)))
            cut = rng.randint(1, max(1, L - 1))
            d = rng.randint(1, 12)
            typ = rng.random()
            if typ < 0.45:
                read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
            elif typ < 0.85:
                read = (contig[p:p + cut] + "".join(rng.choice("ACGT") for _ in range(d)) + contig[p + cut:p + L])[:L]
            else:
                read = contig[p:p + L]
            read = "".join((rng.choice("ACGT") if rng.random() < 0.01 else ch) for ch in read)
            if len(read) < 4:
                read = contig[:4]
            cases.append(dict(anchor=anchor, range_max=Rm, read=read))
        ctx.set_reference([contig.encode()])
        out, bad = T._run_cases(ctx, capi, capi.params(**kw), ob.params(**kw), contig.encode(), cases,
                                dump="gpurun_out/mismatch_soak_%d_%d.txt" % (seed, trial))
        if bad:
            print("g>0 fuzz seed %d trial %d %r: %d of %d differ" % (seed, trial, kw, len(bad), len(cases)), flush=True)
            sys.exit(1)
    print("g>0 fuzz seed %d ok" % seed, flush=True)
ctx.close()
