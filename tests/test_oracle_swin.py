"""Oracle (oracle/swin.py) vs outputs of the reference's own swintransformer.py (tests/golden)."""
import numpy as np
import pytest
import torch

from oracle import swin as O
from tests._recipes import fill_state, swin_param_shapes


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("ws", [7, 12])
def test_window_attention_fwd_bwd(golden, ws):
    g = golden("swin_attn_w%d" % ws)
    p = {"qkv.weight": T(g["qkv_w"]), "qkv.bias": T(g["qkv_b"]), "proj.weight": T(g["proj_w"]),
         "proj.bias": T(g["proj_b"]), "relative_position_bias_table": T(g["table"])}
    for v in p.values():
        v.requires_grad_(True)
    assert torch.equal(O.relative_position_index(ws), T(g["index"]))
    x = T(g["x"]).requires_grad_(True)
    out = O.window_attention(x, T(g["mask"]), p, "", 2)
    torch.testing.assert_close(out, T(g["out"]), atol=2e-6, rtol=1e-5)
    out.backward(T(g["g"]))
    torch.testing.assert_close(x.grad, T(g["dx"]), atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(p["relative_position_bias_table"].grad, T(g["d_table"]), atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(p["qkv.weight"].grad, T(g["d_qkv_w"]), atol=2e-5, rtol=1e-4)
    out2 = O.window_attention(x.detach(), None, p, "", 2)
    torch.testing.assert_close(out2, T(g["out_nomask"]), atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("ws", [7, 12])
def test_basic_layer(golden, ws):
    g = golden("swin_layer_w%d" % ws)
    p = {k[2:]: T(g[k]) for k in g.files if k.startswith("p.")}
    H, W = int(g["H"]), int(g["W"])
    x_out, x_down, wh, ww = O.basic_layer(T(g["x"]), H, W, p, "", 2, 2, ws, True)
    torch.testing.assert_close(x_out, T(g["x_out"]), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(x_down, T(g["x_down"]), atol=1e-5, rtol=1e-5)
    assert (wh, ww) == (int(g["Wh"]), int(g["Ww"]))


def test_full_backbone(golden):
    g = golden("swin_full")
    shapes = swin_param_shapes(32, [2, 2, 2, 2], [1, 2, 4, 8], 7)
    p = fill_state(shapes, int(g["param_seed"]), float(g["param_scale"]))
    cs = float(sum(v.double().abs().sum() for v in p.values()))
    assert abs(cs - float(g["param_checksum"])) < 1e-6 * cs
    outs = O.swin_forward(T(g["img"]), p, 32, [2, 2, 2, 2], [1, 2, 4, 8], 7)
    for k in ("swin1", "swin2", "swin3"):
        torch.testing.assert_close(outs[k], T(g[k]), atol=2e-5, rtol=1e-4)


def test_drop_path_stream(golden):
    g = golden("swin_droppath")
    p = {k[2:]: T(g[k]) for k in g.files if k.startswith("p.")}
    torch.manual_seed(int(g["torch_seed"]))
    y = O.swin_block(T(g["x"]), 7, 7, None, p, "", 2, 7, 0, drop=float(g["rate"]), training=True)
    torch.testing.assert_close(y, T(g["y"]), atol=1e-5, rtol=1e-5)


def test_shift_mask_values():
    m = O.shift_mask(10, 13, 7)
    assert m.shape == (4, 49, 49)
    assert set(m.unique().tolist()) <= {0.0, -100.0}
    assert (m[0] == 0).all()  # top-left window never wraps
