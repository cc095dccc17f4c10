python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_gpu_parity_modules.py -q -m gpu -k "graphed or overfits or early or training_step or rccl or several_batch" 2>&1 | tail -30 > gpurun_out/r06_t15.log
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06_j0.json 2> gpurun_out/r06_j0.err
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --swin T > gpurun_out/r06_j1.json 2> gpurun_out/r06_j1.err
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --size 896 > gpurun_out/r06_j2.json 2> gpurun_out/r06_j2.err
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-roofline --through-loader --loader-shards > gpurun_out/r06_j3.json 2> gpurun_out/r06_j3.err
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-roofline --through-loader > gpurun_out/r06_j4.json 2> gpurun_out/r06_j4.err
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-graphs > gpurun_out/r06_j5.json 2> gpurun_out/r06_j5.err
