#!/bin/bash
# ablations of the head switch inside a run of the attention backward (development)
cd $GRAFT_REPO_ROOT/tools/probes
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
for v in "" "-DABL_NOLDSADD" "-DABL_NOGLOBALFLUSH" "-DABL_NOLDSADD -DABL_NOGLOBALFLUSH"; do
  /opt/rocm/bin/hipcc $F $v attn_bwd_bench.hip ../../divergen_amd/csrc/prof.hip -o /tmp/ab_sw 2>/dev/null
  echo "== $v"; for s in "72 24" "18 48"; do /tmp/ab_sw $s; done
done
