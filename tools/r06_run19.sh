python -m pytest tests/test_gpu_loader.py tests/test_abi_and_host.py -q -m gpu 2>&1 | tail -6 > gpurun_out/r06_t19.log
df -h /dev/shm > gpurun_out/r06_shm.txt
B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-roofline"
$B > gpurun_out/r06_m0.json 2> gpurun_out/r06_m0.err
$B --through-loader --loader-shards > gpurun_out/r06_m1.json 2> gpurun_out/r06_m1.err
$B --through-loader --loader-shards --loader-dev pin=loader > gpurun_out/r06_m2.json 2> gpurun_out/r06_m2.err
$B --through-loader > gpurun_out/r06_m3.json 2> gpurun_out/r06_m3.err
$B --through-loader --loader-dev pin=loader > gpurun_out/r06_m4.json 2> gpurun_out/r06_m4.err
$B > gpurun_out/r06_m5.json 2> gpurun_out/r06_m5.err
bash tools/r06_ab_compact.sh > gpurun_out/r06_ab.log 2>&1
