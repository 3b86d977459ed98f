"""Build libindelminer_amd.so (HIP kernels + C ABI) in-tree for gfx950."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libindelminer_amd.so")
SOURCES = ["im_realign.hip", "im_realign_long.hip", "im_realign_any.hip", "im_results.hip", "im_cluster.hip", "im_depth.hip", "im_support.hip", "im_triage.hip", "im_flush.hip", "im_flushwide.hip", "im_capi.hip", "im_comm.hip"]
HEADERS = [os.path.join(ROOT, "include", "indelminer_amd.h"), os.path.join(CSRC, "im_device.hpp")]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_stamps(verbose=False):
    """Diagnostic library with per-phase cycle stamps in the realign kernel (profiles/ only)."""
    out = os.path.join(HERE, "libindelminer_amd_stamps.so")
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DIM_STAMPS",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-ldl", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


HOST_DIR = os.path.join(HERE, "host")
HOST_BIN = os.path.join(HERE, "indelminer")
HOST_SOURCES = ["imhost.c", "hostio.c", "iminflate.c"]
# imhost.c is one translation unit made of parts it includes in order (the reference-shaped helpers stay static)
HOST_PARTS = ["host_logic.c", "host_setup.c", "host_pipeline.c", "host_multirank.c", "host_walk.c", "host_main.c"]


def build_host(force=False, verbose=False):
    """The C host driver (same CLI as the reference's indelminer), linked against the HIP library."""
    srcs = [os.path.join(HOST_DIR, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST_DIR, q) for q in HOST_PARTS] + [os.path.join(HOST_DIR, "imhost.h"), os.path.join(HOST_DIR, "hostio.h"), os.path.join(HOST_DIR, "iminflate.h"), HEADERS[0], LIB]
    if not force and os.path.exists(HOST_BIN) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_BIN) for d in deps):
        return HOST_BIN
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + HOST_DIR, "-o", HOST_BIN]
    cmd += srcs + ["-L" + HERE, "-lindelminer_amd", "-lz", "-lm", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_BIN


if __name__ == "__main__":
    import sys
    if "--stamps" in sys.argv:
        print(build_stamps(verbose=True))
        sys.exit(0)
    print(build(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
