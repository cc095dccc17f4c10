#!/bin/bash
# development library with the phase clocks of wgrad_lw.hip compiled in: divergen_amd/csrc/_obj/libdgx_dev.so (select with DGX_LIB)
cd "$(dirname "$0")/../divergen_amd/csrc" && python build.py > /dev/null && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -DDGX_GEMM_DEV $WL_EXTRA -c wgrad_lw.hip -o _obj/wgrad_lw_dev.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _obj/libdgx_dev${WL_TAG}.so $(ls _obj/*.hip.o | grep -v "wgrad_lw.hip.o") _obj/wgrad_lw_dev.o && echo built _obj/libdgx_dev${WL_TAG}.so
