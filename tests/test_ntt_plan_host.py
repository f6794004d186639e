"""Host-side proof of the wave-owned NTT tile plan (plonkit_amd/csrc/ntt_plan.h, the index logic of ntt.hip's ntt_pass_w):
tests/host/ntt_plan_check.cpp enumerates every (wave, lane, register) of every round of every shape — coverage, butterfly
pairing, exchange correctness, wave-private regions inside a phase, LDS bank conflicts of the scattered stores.  No GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ntt_plan_host(tmp_path):
    exe = str(tmp_path / "ntt_plan_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "plonkit_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "ntt_plan_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "ntt plan ok" in out.stdout
    # every exchange of every shape is free of bank conflicts on the 16-byte stores
    lines = [l for l in out.stdout.splitlines() if "ds_write_b128" in l]
    assert len(lines) == 14 + 4 + 5 and all("array cycles 8.00" in l for l in lines), out.stdout      # four shapes of the 2048-element tile, two of the 4096-element one
