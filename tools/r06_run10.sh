python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "compact" 2>&1 | tail -5 > gpurun_out/r06_t10.log
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --through-loader --loader-shards --workers 8"
$B --loader-dev nice=10 > gpurun_out/r06_i0.json 2> gpurun_out/r06_i0.err
$B --loader-dev switch_us=200 > gpurun_out/r06_i1.json 2> gpurun_out/r06_i1.err
$B --loader-dev main_threads=2 > gpurun_out/r06_i2.json 2> gpurun_out/r06_i2.err
$B --loader-dev main_threads=8 > gpurun_out/r06_i3.json 2> gpurun_out/r06_i3.err
$B --loader-dev nice=10 --loader-dev main_threads=2 > gpurun_out/r06_i4.json 2> gpurun_out/r06_i4.err
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r06_i5.json 2> gpurun_out/r06_i5.err
