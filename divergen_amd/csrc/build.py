"""Build libdgx.so (all HIP kernels + the C ABI of include/divergen_hip.h) for gfx950.

    python -m divergen_amd.csrc.build        # or __graft_entry__.build()

hipcc cross-compiles without a GPU.  The .so is built IN-TREE (divergen_amd/csrc/libdgx.so) so it
travels with the repo snapshot to the GPU box; it is git-ignored.
Flags: -ffp-contract=off keeps the fp32 op sequence of the index-producing kernels identical to the
CPU oracle (bit-exact RoI geometry / NMS / targets); -munsafe-fp-atomics selects the hardware fp32
atomic add for the ROIAlign / bias-table gradient scatters.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["window_attention.hip", "window_shuffle.hip", "roi_align.hip", "nms_boxes.hip",
           "centernet_targets.hip", "compositor.hip", "optim.hip", "im2col.hip", "wgrad_gemm.hip", "wgrad256.hip", "gemm_nt.hip", "gemm_lw.hip", "gemm_k192.hip", "wgrad_lw.hip", "transpose.hip", "prof.hip", "preprocess.hip", "mask_loss.hip", "layernorm.hip", "residual.hip", "colsum.hip", "groupnorm.hip", "gelu.hip", "detic_loss.hip", "centernet_loss.hip", "cascade_refine.hip", "mask_paste.hip", "grad_bank.hip", "roi_sample.hip", "centernet_decode.hip", "topk_sort.hip", "resnet_ops.hip", "abi.hip"]
OUT = os.path.join(HERE, "libdgx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wno-unused-result"]
if os.environ.get("DGX_DEV") == "1":       # development build: ablation instantiations of the GEMM (tools/gemm_phase_probe.py)
    FLAGS.append("-DDGX_GEMM_DEV")
FLAGS += os.environ.get("DGX_EXTRA_FLAGS", "").split()      # experiments (-D switches of one gpurun call); the product build sets none


def build(verbose=False, force=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    deps = srcs + [os.path.join(HERE, "dgx_common.h"), os.path.join(HERE, "gemm_common.h"),
                   os.path.join(HERE, "..", "..", "include", "divergen_hip.h")]
    objs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "_obj", os.path.basename(s) + ".o")
        objs.append(o)
        hdrs = deps[len(srcs):]
        if force or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    if force or procs or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
