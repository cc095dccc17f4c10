set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in on off; do
  if [ $tag = off ]; then export DGX_GEMM256=0; else unset DGX_GEMM256; fi
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$tag -o p -- python $R/bench.py --dev gemm_log=/tmp/gemm_$tag.log --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > /tmp/bench_$tag.json 2>/tmp/bench_$tag.err
  f=$(find /tmp/p_$tag -name "*kernel_trace.csv" | head -1)
  python $R/tools/gemm_insitu.py $f /tmp/gemm_$tag.log 6 > $R/gpurun_out/r3i/insitu_$tag.txt 2>&1
done
