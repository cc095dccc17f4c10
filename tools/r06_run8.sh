# own top-k / sort of the proposal decode: parity tests, then a same-box A/B of the default line, then the kernels' durations in situ
mkdir -p gpurun_out/r06t
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_modules.py -q -m gpu -k "topk or sort_rows or decode or nms or proposal" 2>&1 | tail -25 > gpurun_out/r06t/tests.log
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
for i in 1 2; do
  timeout 600 $B > gpurun_out/r06t/own_$i.json 2> gpurun_out/r06t/own_$i.err
  timeout 600 $B --no-own-topk > gpurun_out/r06t/torch_$i.json 2> gpurun_out/r06t/torch_$i.err
done
cd /tmp && export TMPDIR=/tmp
for v in own torch; do
  fl=""; [ $v = torch ] && fl="--no-own-topk"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o p -- python $R/bench.py --steps 6 --warmup 4 --no-roofline --no-cpu-baseline $fl > /dev/null 2> $R/gpurun_out/r06t/prof_$v.err
  f=$(find /tmp/p_$v -name "*kernel_stats.csv" | head -1)
  grep -i "topk\|sort\|radix\|gather_boxes\|cn_scores\|cn_decode\|nms\|bitonic\|segmented\|index_rows" $f > $R/gpurun_out/r06t/kern_$v.csv
  python $R/tools/decode_window.py $(find /tmp/p_$v -name "*kernel_trace.csv" | head -1) 2 > $R/gpurun_out/r06t/window_$v.txt 2>&1
done
