"""indelminer_amd/host/iminflate.c (the BGZF blocks' raw DEFLATE decoder) against zlib: identical output on valid streams of
every kind zlib writes, -1 on truncated ones, and memory safety on corrupted ones -- the harness allocates every buffer at
exactly the size the decoder's contract lets it touch and is built with AddressSanitizer + UBSan (CPU only)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _harness(tmp_path, sanitize=True):
    exe = str(tmp_path / "inflate_fuzz")
    cmd = ["gcc", "-O1", "-g", "-std=gnu11", "-I" + os.path.join(ROOT, "indelminer_amd", "host"), "-o", exe,
           os.path.join(ROOT, "tests", "support", "inflate_fuzz.c"), os.path.join(ROOT, "indelminer_amd", "host", "iminflate.c"), "-lz"]
    if sanitize:
        cmd[1:1] = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    subprocess.check_call(cmd)
    return exe


def test_inflate_matches_zlib_and_survives_bad_streams(tmp_path):
    exe = _harness(tmp_path)
    # the decoder has two instances of its loop (with and without BMI2's bit-field instructions, picked at run time): both
    for seed, plain in ((1, "0"), (2, "0"), (1, "1"), (3, "1")):
        r = subprocess.run([exe, "350", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", INDELMINER_INFLATE_PLAIN=plain))
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        got = json.loads(r.stdout.decode())
        assert got["valid"] == 350 and got["truncated"] > 1500 and got["corrupted"] > 3000
        assert got["corrupted_rejected"] > got["corrupted"] // 4          # most flips break a code or a length; none may crash
