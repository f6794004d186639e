"""Host field arithmetic around the kernels (plonkit_amd/csrc/hostmath.h): tests/host/hostmath_check.cpp checks the unrolled Montgomery
product against the looped form it replaced, the division-step inverse of round 6 against the binary Euclid it replaced and the Fermat exponentiation (and that it never needs its fallback), and the shortened
square-and-multiply, on random and extreme values of Fr and Fq.  No GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hostmath_host(tmp_path):
    exe = str(tmp_path / "hostmath_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "plonkit_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "hostmath_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "hostmath ok" in out.stdout and "Fr: 0 mismatches; 0 fallbacks" in out.stdout and "Fq: 0 mismatches; 0 fallbacks" in out.stdout
