"""Host verifier (plk_verify / plk_pairing_check, pure CPU): plonk::verify of the reference
(src/plonk.rs:189-210, test_verify src/tests.rs:76-81) on the committed golden vk.bin / proof.bin, with a
real BN254 pairing — the oracle's verifier checks the last step through the tau = 42 trapdoor instead, so the
two are independent implementations of the same predicate."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle_lib as ol
from oracle import plonk_oracle as po

R_MOD = po.R_MOD


@pytest.fixture(scope="module")
def vk_proof(golden_dir):
    return (open(os.path.join(golden_dir, "vk.bin"), "rb").read(), open(os.path.join(golden_dir, "proof.bin"), "rb").read())


def test_golden_proof_verifies(vk_proof):
    import plonkit_amd as pa
    vk, proof = vk_proof
    assert pa.verify(vk, proof)
    assert po.verify(po.read_vk(vk), po.read_proof(proof))          # the oracle agrees


def _with(proof, **changes):
    P = po.read_proof(proof)
    for k, v in changes.items():
        setattr(P, k, v(getattr(P, k)))
    return po.write_proof(P)


def test_every_tampered_field_is_rejected(vk_proof):
    """each scalar of the proof moved by one, each commitment replaced by another valid curve point:
    the verifier and the oracle's verifier must both say no (the commitment cases reach the pairing)."""
    import plonkit_amd as pa
    vk, proof = vk_proof
    bump = lambda x: (x + 1) % R_MOD
    bump_first = lambda xs: [bump(xs[0])] + list(xs[1:])
    swap01 = lambda xs: [xs[1], xs[0]] + list(xs[2:])
    P0 = po.read_proof(proof)
    cases = {
        "inputs": bump_first, "wire_values_at_z": bump_first, "wire_values_at_z_omega": bump_first,
        "permutation_polynomials_at_z": bump_first, "grand_product_at_z_omega": bump,
        "quotient_polynomial_at_z": bump, "linearization_polynomial_at_z": bump,
        "wire_commitments": swap01, "quotient_poly_commitments": swap01,
        "grand_product_commitment": lambda c: P0.wire_commitments[0],
        "opening_at_z_proof": lambda c: P0.opening_at_z_omega_proof,
        "opening_at_z_omega_proof": lambda c: P0.opening_at_z_proof,
    }
    for field, change in cases.items():
        bad = _with(proof, **{field: change})
        assert bad != proof, field
        assert not pa.verify(vk, bad), field
        assert not po.verify(po.read_vk(vk), po.read_proof(bad)), field


def test_wrong_key_and_malformed_files(vk_proof):
    import plonkit_amd as pa
    vk, proof = vk_proof
    V = po.read_vk(vk)
    V.permutation_commitments = [V.permutation_commitments[1], V.permutation_commitments[0]] + list(V.permutation_commitments[2:])
    assert not pa.verify(po.write_vk(V), proof)
    for bad_vk, bad_proof in ((vk[:-1], proof), (vk, proof[:-1]), (vk, proof + b"\0"), (b"", proof)):
        with pytest.raises(pa.PlkError):
            pa.verify(bad_vk, bad_proof)
    off_curve = bytearray(proof); off_curve[-1] ^= 1                 # W_zw no longer on the curve
    with pytest.raises(pa.PlkError):
        pa.verify(vk, bytes(off_curve))
    big = bytearray(proof); big[16:48] = (R_MOD).to_bytes(32, "big")  # public input == r: not canonical
    with pytest.raises(pa.PlkError):
        pa.verify(vk, bytes(big))
    g2_swapped = vk[:-256] + vk[-128:] + vk[-256:-128]               # e(A, 42 g2) e(B, g2) is not 1
    assert not pa.verify(g2_swapped, proof)


def test_pairing_bilinearity_against_the_tau42_key(golden_crs):
    """e(42^k P, G2) * e(-42^(k-1) P, 42 G2) == 1 for the points of the golden 2^10 key, and not for neighbours"""
    import plonkit_amd as pa
    g2 = pa.crs42_g2_bytes()
    assert g2 == golden_crs.g2_raw
    q0, q1 = g2[:128], g2[128:]
    for k in (1, 2, 513, 1023):
        hi, lo = golden_crs.g1[k], golden_crs.g1[k - 1]
        assert pa.pairing_check(hi, q0, ol.g1_neg(lo), q1)
        assert not pa.pairing_check(hi, q0, lo, q1)
        assert not pa.pairing_check(golden_crs.g1[k - 1], q0, ol.g1_neg(lo), q1)
    inf = np.zeros(8, dtype=np.uint64)
    assert pa.pairing_check(inf, q0, inf, q1)                         # e(O, .) = 1
    assert not pa.pairing_check(golden_crs.g1[1], q0, inf, q1)
    with pytest.raises(pa.PlkError):
        pa.pairing_check(golden_crs.g1[1], q0[:-1] + bytes([q0[-1] ^ 1]), inf, q1)   # not on the twist


def test_cli_verify_exit_codes(vk_proof, golden_dir, tmp_path):
    """`plonkit verify -p -v`: 0 when valid; 400 (the shell sees 144) when invalid, as src/bin/main.rs:432-437"""
    import plonkit_amd as pa
    cli = os.path.join(os.path.dirname(pa.lib_path()), "plonkit")
    vk, proof = vk_proof
    vkp, pp, badp = str(tmp_path / "vk.bin"), str(tmp_path / "proof.bin"), str(tmp_path / "bad.bin")
    open(vkp, "wb").write(vk); open(pp, "wb").write(proof)
    open(badp, "wb").write(_with(proof, quotient_polynomial_at_z=lambda x: (x + 1) % R_MOD))
    assert subprocess.call([cli, "verify", "-p", pp, "-v", vkp], stderr=subprocess.DEVNULL) == 0
    assert subprocess.call([cli, "verify", "-p", pp, "-v", vkp, "-t", "keccak"], stderr=subprocess.DEVNULL) == 0
    assert subprocess.call([cli, "verify", "-p", badp, "-v", vkp], stderr=subprocess.DEVNULL) == 144
    assert subprocess.call([cli, "verify", "-p", pp, "-v", vkp, "-t", "rescue"], stderr=subprocess.DEVNULL) == 101
    assert subprocess.call([cli, "verify", "-p", str(tmp_path / "missing.bin"), "-v", vkp], stderr=subprocess.DEVNULL) == 101
