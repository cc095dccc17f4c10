"""CPU: divergen_amd/csrc/winmap.h (the compact window order shared by the LayerNorm, GEMM-epilogue and attention kernels) compiled
for the host and checked against a brute-force enumeration of the reference's pad -> roll -> partition
(swintransformer.py:216-233): real / padding classification, compact row of every token, its inverse, the padding rows' order."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_compact_window_map_against_brute_force(tmp_path):
    exe = str(tmp_path / "winmap_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "winmap_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
    assert int(out.stdout.split()[1]) >= 30
