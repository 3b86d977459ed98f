"""Builds tests/shim/indelminer_shim: the product's host driver linked against tests/shim/im_shim.c, which implements the C ABI
on the CPU with the oracle -- TEST INFRASTRUCTURE (host-logic tests, multi-rank tests, and bench.py's cpu_baseline legs, where it
is the reference's path without its per-candidate strlen of the contig).  Never part of the product."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHIM = os.path.join(ROOT, "tests", "shim", "indelminer_shim")


def build_shim():
    srcs = [os.path.join(ROOT, "indelminer_amd", "host", "imhost.c"), os.path.join(ROOT, "indelminer_amd", "host", "hostio.c"),
            os.path.join(ROOT, "indelminer_amd", "host", "iminflate.c"),
            os.path.join(ROOT, "tests", "shim", "im_shim.c"), os.path.join(ROOT, "oracle", "im_oracle.c"),
            os.path.join(ROOT, "oracle", "im_oracle_triage.c")]
    from indelminer_amd import build
    parts = [os.path.join(ROOT, "indelminer_amd", "host", q) for q in build.HOST_PARTS]      # included by imhost.c
    parts += [os.path.join(ROOT, "indelminer_amd", "host", h) for h in ("imhost.h", "hostio.h", "iminflate.h")]
    parts += [os.path.join(ROOT, "include", "indelminer_amd.h"), os.path.join(ROOT, "oracle", "im_oracle.h")]
    if os.path.exists(SHIM) and all(os.path.getmtime(s) <= os.path.getmtime(SHIM) for s in srcs + parts):
        return SHIM
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "indelminer_amd", "host"), "-o", SHIM] + srcs + ["-lz", "-lm"])
    return SHIM
