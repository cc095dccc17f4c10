#!/bin/bash
# In-situ time per GEMM shape (rocprofv3 kernel trace joined with the library's launch log) for a bench configuration:
#   bash tools/r05_insitu.sh <tag> "<bench args>" "<dev settings: e.g. gemm_lw=1>"...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
tag=$1; args=$2; shift 2
for v in "default" "$@"; do
  dev=""; [ "$v" != "default" ] && for kv in $v; do dev="$dev --dev $kv"; done
  n=$(echo "$v" | tr ' =' '__')
  rm -rf /tmp/p_$n
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$n -o p -- python $R/bench.py $args $dev --dev gemm_log=/tmp/g_$n.log --no-graphs --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > $O/bench_${tag}_$n.json 2> $O/trace_$n.err
  python $R/tools/gemm_insitu.py $(find /tmp/p_$n -name "*kernel_trace.csv" | head -1) /tmp/g_$n.log 9 > $O/insitu_${tag}_$n.txt 2>&1
  echo "== $tag $v: $(tail -1 $O/insitu_${tag}_$n.txt)"
done
