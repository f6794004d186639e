"""plonkit_amd — MI355X-native PLONK prover hot path (BN254 NTT / coset NTT / Pippenger G1 MSM /
KZG commit / prover rounds) behind a C ABI (include/plonkit_amd.h, lib/libplonkit_amd.so).

This Python package is only the thin host-side mirror used by tests, bench.py and the multi-GPU
driver: ctypes bindings + torch plumbing (device memory, streams, torch.distributed).  There is no
CPU fallback: creating a Context without a gfx950 device raises.
"""
import os as _os

# up to three commitments may be in flight on three HIP streams (MSM slots); give them distinct hardware queues.  Only effective
# when set before the HIP runtime initialises, i.e. before the first GPU call of the process (import time is fine).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import (PlkError, lib, lib_path, Context, last_error, have_gpu,   # noqa: F401
                   g1_sum_jacobian, g1_to_bytes, g1_from_bytes, fr_to_bytes, fr_from_bytes,
                   Transcript, keccak256, Circuit, SetupForProver, verify, pairing_check, crs42_g2_bytes,
                   comm_unique_id)

__all__ = ["PlkError", "lib", "lib_path", "Context", "last_error", "have_gpu", "g1_sum_jacobian",
           "g1_to_bytes", "g1_from_bytes", "fr_to_bytes", "fr_from_bytes", "Transcript", "keccak256", "Circuit", "SetupForProver",
           "verify", "pairing_check", "crs42_g2_bytes", "comm_unique_id"]
