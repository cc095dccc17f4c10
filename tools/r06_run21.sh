# what a batch size WITHOUT captured head segments costs (the shipped SCALE_RANGE (0.1, 2.0) produces them: ~1 in 5 batches)
mkdir -p gpurun_out/r06c
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
for s in 1024 768 512; do
  timeout 600 $B --size $s > gpurun_out/r06c/g_$s.json 2> gpurun_out/r06c/g_$s.err
  timeout 600 $B --size $s --no-graphs > gpurun_out/r06c/e_$s.json 2> gpurun_out/r06c/e_$s.err
done
timeout 900 python bench.py --through-loader --loader-shards --loader-scale-range 0.1 2.0 --steps 400 --warmup 100 --no-cpu-baseline --no-roofline > gpurun_out/r06c/loader_scale_400.json 2> gpurun_out/r06c/loader_scale_400.err
