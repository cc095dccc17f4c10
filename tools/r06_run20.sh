python -m pytest tests/test_gpu_ddp.py tests/test_gpu_model.py -q -m gpu -k "rccl or graphed or several" 2>&1 | tail -5 > gpurun_out/r06_t20.log
B="python bench.py --steps 50 --warmup 12 --no-cpu-baseline"
$B > gpurun_out/r06_n0.json 2> gpurun_out/r06_n0.err
$B --no-roofline --through-loader --loader-shards > gpurun_out/r06_n1.json 2> gpurun_out/r06_n1.err
$B --no-roofline --through-loader --loader-shards --loader-dev pin=loader > gpurun_out/r06_n2.json 2> gpurun_out/r06_n2.err
$B --no-roofline --through-loader > gpurun_out/r06_n3.json 2> gpurun_out/r06_n3.err
$B --no-roofline --through-loader --loader-dev pin=loader > gpurun_out/r06_n4.json 2> gpurun_out/r06_n4.err
$B > gpurun_out/r06_n5.json 2> gpurun_out/r06_n5.err
$B --swin T > gpurun_out/r06_n6.json 2> gpurun_out/r06_n6.err
