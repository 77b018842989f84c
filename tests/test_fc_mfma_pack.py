"""The MFMA fragment blob of the cost-query MLP against a host emulation of the kernel's tile arithmetic (no GPU)."""
import os
import shutil
import subprocess

import pytest

import common


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_fc_mfma_blob_reproduces_the_mlp_through_the_kernels_tile_arithmetic(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "fc_mfma_emu")
    src = os.path.join(common.ROOT, "tests", "cpp", "fc_mfma_emu.hip")
    inc = os.path.join(common.ROOT, "art_planner_amd", "csrc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-ffp-contract=off", "-I" + inc,
                           "-I" + os.path.join(common.ROOT, "include"), "-o", exe, src, "-ldl"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "worst |emulated kernel - reference|" in out.stdout
