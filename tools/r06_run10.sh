# needs profiles/r06_decode_side_stream.patch applied (the side-stream decode was measured and not kept)
# side-stream decode + NMS cap stop: tests, same-box A/B, timeline
mkdir -p gpurun_out/r06v
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_parity_modules.py -q -m gpu -x -k "side_stream or nms or decode or early or sampler or training_step or topk or sort_rows" 2>&1 | tail -25 > gpurun_out/r06v/tests.log
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  timeout 600 $B > gpurun_out/r06v/side_$i.json 2> gpurun_out/r06v/side_$i.err
  timeout 600 $B --no-decode-side-stream > gpurun_out/r06v/one_$i.json 2> gpurun_out/r06v/one_$i.err
done
timeout 600 $B --through-loader --loader-shards --workers 12 > gpurun_out/r06v/loader_side.json 2> gpurun_out/r06v/loader_side.err
cd /tmp && export TMPDIR=/tmp
for v in side one; do
  fl=""; [ $v = one ] && fl="--no-decode-side-stream"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o p -- python $R/bench.py --steps 6 --warmup 4 --no-roofline --no-cpu-baseline $fl > /dev/null 2> $R/gpurun_out/r06v/prof_$v.err
  python $R/tools/decode_window.py $(find /tmp/p_$v -name "*kernel_trace.csv" | head -1) 2 > $R/gpurun_out/r06v/window_$v.txt 2>&1
done
