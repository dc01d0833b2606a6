"""Bitwise micro-checks of hand-written wave primitives on the GPU (compiled with hipcc at test time)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_dpp_permlane_butterfly_equals_shfl_xor_bitwise(tmp_path):
    """k5_edge_score sums the 64 hidden units with an xor butterfly (strides 32..1, the oracle's order) built from
    DPP moves and the gfx950 permlane swaps; it must pair the same lanes as __shfl_xor at every stride and give
    the same bits as the __shfl_xor butterfly on random fp32 data."""
    exe = str(tmp_path / "butterfly_check")
    src = os.path.join(HERE, "micro", "butterfly_check.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-o", exe, src])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
