python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_modules.py tests/test_gpu_loader.py -q -m gpu -k "compact or fused_swin_block or layernorm or window or loader or instpool or do_train or chain" 2>&1 | tail -25 > gpurun_out/r06_t9.log
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06_h0.json 2> gpurun_out/r06_h0.err
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06_h1.json 2> gpurun_out/r06_h1.err
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --through-loader --loader-shards"
$B --workers 8 > gpurun_out/r06_h2.json 2> gpurun_out/r06_h2.err
$B --workers 8 > gpurun_out/r06_h3.json 2> gpurun_out/r06_h3.err
$B --workers 16 > gpurun_out/r06_h4.json 2> gpurun_out/r06_h4.err
