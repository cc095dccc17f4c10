B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --through-loader --loader-shards"
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r06_f0.json 2> gpurun_out/r06_f0.err
$B --workers 8 > gpurun_out/r06_f1.json 2> gpurun_out/r06_f1.err
$B --workers 8 --loader-dev main_threads=1 > gpurun_out/r06_f2.json 2> gpurun_out/r06_f2.err
$B --workers 12 > gpurun_out/r06_f3.json 2> gpurun_out/r06_f3.err
$B --workers 16 > gpurun_out/r06_f4.json 2> gpurun_out/r06_f4.err
$B --workers 6 --loader-dev main_threads=2 > gpurun_out/r06_f5.json 2> gpurun_out/r06_f5.err
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --through-loader --workers 12 > gpurun_out/r06_f6.json 2> gpurun_out/r06_f6.err
python -m pytest tests/test_gpu_loader.py tests/test_gpu_model.py -q -m gpu -k "do_train or vs_assembled_oracle" 2>&1 | tail -15 > gpurun_out/r06_t7.log
