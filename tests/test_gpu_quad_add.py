"""-m gpu: the distributed four-lane addition (plonkit_amd/csrc/ec29_quad_dev.h) against the lane-wise XYZZ addition on the device: random pairs,
doubling, opposite points, infinity on either side, and chains of 24 dependent additions fed back in the distributed form (tests/host/quad_add_check.hip).
The reduction kernels built on it (msm_small_fold / _planes, msm_window_sums_quad, msm_task_reduce_quad, the in-quad sums of the two accumulate kernels)
are pinned end to end by tests/test_gpu_kernels.py against the oracle and the tau = 42 trapdoor."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_distributed_quad_addition_equals_the_lane_wise_one(tmp_path):
    exe = str(tmp_path / "quad_add_check")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "plonkit_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "quad_add_check.hip"), "-o", exe], stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "quad_add: 0 mismatches of 16384" in r.stdout, r.stdout
    seen = [int(x) for x in r.stdout.splitlines()[0].replace("special cases seen:", "").split()[1::2]]
    assert min(seen) >= 2000, r.stdout
