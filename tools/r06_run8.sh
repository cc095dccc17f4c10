python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_modules.py -q -m gpu -x -k "compact or fused_swin_block or layernorm or window" 2>&1 | tail -25 > gpurun_out/r06_t8.log
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06_g0.json 2> gpurun_out/r06_g0.err
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --size 896 > gpurun_out/r06_g1.json 2> gpurun_out/r06_g1.err
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --through-loader --loader-shards"
$B --workers 8 > gpurun_out/r06_g2.json 2> gpurun_out/r06_g2.err
$B --workers 8 --loader-dev affinity=2 > gpurun_out/r06_g3.json 2> gpurun_out/r06_g3.err
$B --workers 5 > gpurun_out/r06_g4.json 2> gpurun_out/r06_g4.err
