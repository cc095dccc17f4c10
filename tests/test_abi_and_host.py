"""CPU-only checks: the C-ABI library loads and exports every symbol include/*.h declares; host-side
integer logic (shift regions) matches the oracle; ops refuse CPU tensors loudly (no fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "divergen_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dgx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from divergen_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from divergen_amd.csrc.build import build
        build()
    L = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(L, s), "libdgx.so does not export %s" % s
        assert s in _lib.SIGNATURES, "ctypes signature missing for %s" % s
    assert set(_lib.SIGNATURES) == set(syms)
    assert L.dgx_build_arch() == b"gfx950"


def test_shift_regions_match_oracle_mask():
    from divergen_amd.layers import shift_regions
    from oracle import swin as O
    for (H, W, ws) in [(10, 13, 7), (14, 25, 12), (7, 7, 7), (24, 36, 12)]:
        reg = shift_regions(H, W, ws).float()
        mask = (reg[:, None, :] != reg[:, :, None]).float() * -100.0
        assert torch.equal(mask, O.shift_mask(H, W, ws))


def test_ops_fail_loudly_on_cpu_tensors():
    from divergen_amd import _lib
    from divergen_amd import layers as la
    with pytest.raises(_lib.DgxError):
        la.window_gather(torch.zeros(1, 49, 32), 7, 7, 7, 0)
    with pytest.raises(_lib.DgxError):
        la.nms(torch.zeros(4, 4), torch.zeros(4), 0.5)
    with pytest.raises(_lib.DgxError):
        la.window_attention_core(torch.zeros(1, 49, 96, dtype=torch.bfloat16), torch.zeros(169, 1), None, 1, 1, 7, 1.0)
