"""The lazy 9 x 29-bit field layer and the XYZZ group law built on it (plonkit_amd/csrc/field29_dev.h, ec29_dev.h) compiled
for the HOST and compared with the 8 x 32-bit layer on random inputs: products, squarings, fused sums, lazy add/sub
chains, zero tests, the quotient-estimate reduction, and a random walk of mixed additions / doublings / full additions
including P + P and P - P.  No GPU involved (hipcc only compiles); the GPU suite pins both layers to the oracle."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_field29_and_ec29_against_the_32_bit_layer(tmp_path):
    exe = str(tmp_path / "field29_check")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "plonkit_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "field29_check.hip"), "-o", exe], stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert "field29: 0 mismatches" in r.stdout and "ec29: 0 mismatches" in r.stdout and "glv: 0 mismatches" in r.stdout
    # known-answer lines: canonical a, b and the product computed by the W layer, checked with Python integers
    mod = {"Fr": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
           "Fq": 21888242871839275222246405745257275088696311157297823662689037894645226208583}
    kats = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("KAT ")]
    assert len(kats) == 80
    for _, field, a, b, prod in kats:
        assert int(a, 16) * int(b, 16) % mod[field] == int(prod, 16), (field, a, b)
    # GLV split known answers (glv_dev.h: the scalar multiplication of the G1 iNTT): k = k1 + k2 * lambda (mod r), halves < 2^128
    lam = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
    assert pow(lam, 3, mod["Fr"]) == 1 and lam != 1
    glv = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("GLV ")]
    assert len(glv) == 40
    for _, k, k1, n1, k2, n2 in glv:
        k, k1, k2 = int(k, 16), int(k1, 16), int(k2, 16)
        assert k1 < (1 << 128) and k2 < (1 << 128)
        assert ((-k1 if n1 == "1" else k1) + (-k2 if n2 == "1" else k2) * lam - k) % mod["Fr"] == 0, hex(k)
    assert int(glv[1][1], 16) == mod["Fr"] - 1 and int(glv[0][1], 16) == 0
