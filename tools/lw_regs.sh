#!/bin/bash
# register / spill report of the loader-wave GEMM instantiations (runs without a GPU)
cd "$(dirname "$0")/../divergen_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Rpass-analysis=kernel-resource-usage -c gemm_lw.hip -o /tmp/gemm_lw.o 2>&1 | grep -E "Function Name|VGPRs:|Spill|ScratchSize|error|warning: " | paste - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/gemm_lw.hip:[0-9]*:[0-9]*: remark: //g; s/Function Name: _Z14gemm_lw_kernelI//; s/EvN7dgxgemm5GemmPE//'
