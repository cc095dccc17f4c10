#!/bin/bash
# ablations of phase 1 of the attention backward in the standalone harness (wrong results, timing only): what is the phase bound by?
cd $GRAFT_REPO_ROOT/tools/probes
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DDIAG_CLOCK -DDIAG_WAVE=1"
for v in "" "-DABL_EXP" "-DABL_META" "-DABL_DSWRITE" "-DABL_TR" "-DABL_MFMA1" "-DABL_EXP -DABL_META -DABL_DSWRITE -DABL_TR"; do
  /opt/rocm/bin/hipcc $F $v attn_bwd_bench.hip ../../divergen_amd/csrc/prof.hip -o /tmp/ab 2>&1 | grep -E "error" | head -3
  echo "[$v]"; /tmp/ab 968 6 | grep -E "bwd|phase1|sync2|phase2"
done
