python -m pytest tests/test_gpu_loader.py tests/test_gpu_model.py -q -m gpu -k "do_train or several_batch_sizes or early" 2>&1 | tail -40 > gpurun_out/r06_t3.log
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > gpurun_out/r06_c1.json 2> gpurun_out/r06_c1.err
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --through-loader --loader-shards > gpurun_out/r06_c2.json 2> gpurun_out/r06_c2.err
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --through-loader --loader-shards --workers 4 > gpurun_out/r06_c3.json 2> gpurun_out/r06_c3.err
python bench.py --steps 30 --warmup 12 --no-cpu-baseline --no-roofline --through-loader --loader-scale-range 0.1 2.0 > gpurun_out/r06_c4.json 2> gpurun_out/r06_c4.err
