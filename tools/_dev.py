"""Development knobs of libdgx for the probes under tools/: the library reads no environment variable (include/divergen_hip.h,
dgx_dev_set), so the probes translate the variables their shell scripts set into setter calls."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

ENV = {"DGX_GEMM_LW": "gemm_lw", "DGX_GEMM_2WG": "gemm_2wg", "DGX_GEMM_SPLITK": "gemm_splitk", "DGX_WGRAD_LW": "wgrad_lw"}


def dev_set(key, value):
    from divergen_amd import _lib
    rc = _lib.lib().dgx_dev_set(key.encode(), int(value))
    assert rc == 0, (key, value)


def set_tile(tile):
    """'256x192' or None."""
    if tile:
        bm, bn = tile.split("x")
        dev_set("gemm_tile", int(bm) * 1000 + int(bn))
    else:
        dev_set("gemm_tile", 0)


def apply_env():
    from divergen_amd import _lib
    for e, k in ENV.items():
        if e in os.environ:
            dev_set(k, os.environ[e])
    if os.environ.get("DGX_GEMM_TILE"):
        set_tile(os.environ["DGX_GEMM_TILE"])
    if os.environ.get("DGX_GEMM_LOG"):
        assert _lib.lib().dgx_dev_gemm_log(os.environ["DGX_GEMM_LOG"].encode()) == 0
