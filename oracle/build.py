"""Build recipe for the oracle's native pieces (TEST INFRASTRUCTURE, see oracle/__init__.py).

  build_oracle()  gcc  oracle/roi_nms.c           -> oracle/_build/liboracle.so   (always)
  build_ref()     g++  reference in-tree CPU sources (ROIAlignRotated_cpu.cpp, nms_rotated_cpu.cpp, cocoeval.cpp),
                  compiled where they lie under /root/reference, + oracle/ref_shim.cpp -> oracle/_ref/dgref.so
                  (only when /root/reference exists, i.e. in the authoring container;
                  the GPU box uses the prebuilt file that travels with the snapshot)

Building the checker is not using it: only tests/, smoke() and bench.py's cpu_baseline load these.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CSRC = "/root/reference/BSGAL/third_party/CenterNet2/detectron2/layers/csrc"


def _stale(out, srcs):
    return (not os.path.exists(out)) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def build_oracle(verbose=False):
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "liboracle.so")
    src = os.path.join(HERE, "roi_nms.c")
    if _stale(out, [src]):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", out, src, "-lm"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_ref(verbose=False):
    """Returns the path of oracle/_ref/dgref.so, building it if the reference is present."""
    out_dir = os.path.join(HERE, "_ref")
    out = os.path.join(out_dir, "dgref.so")
    if not os.path.isdir(REF_CSRC):
        return out if os.path.exists(out) else None
    os.makedirs(out_dir, exist_ok=True)
    srcs = [os.path.join(HERE, "ref_shim.cpp"),
            os.path.join(REF_CSRC, "ROIAlignRotated", "ROIAlignRotated_cpu.cpp"),
            os.path.join(REF_CSRC, "nms_rotated", "nms_rotated_cpu.cpp"),
            os.path.join(REF_CSRC, "cocoeval", "cocoeval.cpp")]
    if not _stale(out, srcs):
        return out
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], REF_CSRC]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DTORCH_EXTENSION_NAME=dgref",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + i for i in inc] + srcs
    cmd += ["-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-Wl,-rpath," + libdir, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def load_ref():
    """Import oracle/_ref/dgref.so as a python module (needs torch imported first)."""
    import importlib.util
    import torch  # noqa: F401
    path = build_ref()
    if path is None or not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("dgref", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build_oracle(verbose=True))
    print(build_ref(verbose=True))
