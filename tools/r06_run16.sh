python -m pytest tests/test_gpu_ddp.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_t16.log
python bench.py --steps 40 --warmup 10 > gpurun_out/r06_k0.json 2> gpurun_out/r06_k0.err
python bench.py --gpus 2 --backend gloo --share-device 0 --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r06_k1.json 2> gpurun_out/r06_k1.err
python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_full3.log
