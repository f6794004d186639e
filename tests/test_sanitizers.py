"""AddressSanitizer + UndefinedBehaviorSanitizer pass over the HOST half of the library (SURVEY.md §5 / VERDICT r1: "no
sanitizer run on the host C++"): loaders, transpiler, key codec, transcript and the verifier with its pairing are compiled by
gcc with -fsanitize=address,undefined (tests/host/sanitize_host.cpp pulls the four translation units in) and fed the golden
files of the reference plus a few thousand random mutations of each.  Every call must come back with a status: no
out-of-bounds access, no signed overflow, no misaligned or null access, no exception through the extern "C" boundary —
and no mutated (vk, proof) pair may verify.  No GPU involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not on PATH")
def test_host_half_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
                           os.path.join(ROOT, "tests", "host", "sanitize_host.cpp"), "-o", exe])
    files = [os.path.join(GOLD, f) for f in ("r1cs_sample.bin", "circuit.r1cs.json", "witness.json", "vk.bin", "proof.bin", "setup_2pow10.key")]
    r = subprocess.run([exe] + files + ["400"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "0 forged" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, (r.stdout + r.stderr)[-4000:]
