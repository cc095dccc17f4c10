#!/bin/bash
# Round-4 evidence for the weight-gradient kernels: standalone groups (loader-wave form vs split-M form), phase clocks of the loader-wave
# kernel (development build: tools/build_dev_wl.sh), PMC passes, and the three micro-probes its design rests on.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wlw; mkdir -p $O; cd $R
{ echo "# PROBE_BETA=0 python tools/wgrad_lw_probe.py 7 4   (loader-wave form where the library takes it; bias sums on)"; PROBE_BETA=0 python tools/wgrad_lw_probe.py 7 4 2>&1 | grep stage
  echo "# DGX_WGRAD_LW=0 ...   (split-M 256x256 form, launches of <= 12 problems)"; PROBE_BETA=0 DGX_WGRAD_LW=0 python tools/wgrad_lw_probe.py 7 4 2>&1 | grep stage
  echo "# PROBE_BIAS=0 ...   (no bias sums), loader-wave / split-M"; PROBE_BIAS=0 PROBE_BETA=0 python tools/wgrad_lw_probe.py 7 4 2>&1 | grep stage; PROBE_BIAS=0 PROBE_BETA=0 DGX_WGRAD_LW=0 python tools/wgrad_lw_probe.py 7 4 2>&1 | grep stage
} > $O/r04_wgrad_lw_probe.txt
{ echo "# PROBE_BETA=0 PROBE_BIAS=0 python tools/wgrad_lw_clocks.py   (workgroup 0, stamps by s_memtime; stamps cost ~140 cycles each)"; PROBE_BETA=0 PROBE_BIAS=0 python tools/wgrad_lw_clocks.py 2>&1 | grep -v amdgpu
  echo "# PROBE_BIAS=1"; PROBE_BETA=0 PROBE_BIAS=1 python tools/wgrad_lw_clocks.py 2>&1 | grep -v amdgpu
  echo "# DGX_WGRAD_LW_DIAG=1: every workgroup streams the panels of item 0 (L2-resident operands)"; DGX_WGRAD_LW_DIAG=1 PROBE_BETA=0 PROBE_BIAS=0 python tools/wgrad_lw_clocks.py 2>&1 | grep -v amdgpu
} > $O/r04_wgrad_lw_clocks.txt
tools/probes/mfma_rate_probe > $O/r04_mfma_rate_probe.txt 2>&1
tools/probes/mfma_mix_probe > $O/r04_mfma_mix_probe.txt 2>&1
tools/probes/soffset_probe > $O/r04_soffset_probe.txt 2>&1
PROBE_BETA=0 bash tools/r04_wgrad_lw_pmc.sh > $O/r04_wgrad_lw_pmc.txt 2>&1
tail -3 $O/r04_wgrad_lw_pmc.txt
