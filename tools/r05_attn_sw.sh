#!/bin/bash
# what the attention backward's bias-table gradient costs a launch: the kernel as built against a build that skips the reduction altogether
# (-DABL_NOFLUSH; that build also loses the 36 accumulations per window of phase 1, so it overstates the flush: 53 vs 44 us at 72 x 24,
# of which the slot hand-over is 1.5 us by -DABL_NOCOUNT; a two-pass form of the per-entry sums with independent reads changed nothing).  The forms this replaced -- 36 ds_add_f32 per lane into an LDS table + 529 global float atomics per workgroup (rounds 2-4:
# +24-27 us per launch), a device-scope fence in front of the last-arriver counter (+40 us per flush) -- are in the history of
# divergen_amd/csrc/window_attention.hip before "Attention backward: bias-table gradient without LDS float atomics"; their timings are in
# profiles/r05_attn_bwd_phases.txt.
cd $GRAFT_REPO_ROOT/tools/probes
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
for v in "" "-DABL_NOCOUNT" "-DABL_NOFLUSH"; do
  /opt/rocm/bin/hipcc $F $v attn_bwd_bench.hip ../../divergen_amd/csrc/prof.hip -o /tmp/ab_sw 2>/dev/null
  echo "== ${v:-as built}"; for s in "968 6" "242 12" "72 24" "18 48"; do /tmp/ab_sw $s; done
done
