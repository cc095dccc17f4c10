#!/bin/bash
# builds the standalone attention timing harness: plain + clock-instrumented (wave 0)
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
/opt/rocm/bin/hipcc $F $EXTRA -DDIAG_CLOCK -DDIAG_WAVE=0 attn_bwd_bench.hip ../../divergen_amd/csrc/prof.hip -o attn_bwd_bench_w0
/opt/rocm/bin/hipcc $F $EXTRA attn_bwd_bench.hip ../../divergen_amd/csrc/prof.hip -o attn_bwd_bench_w8
