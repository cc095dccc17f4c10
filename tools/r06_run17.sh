# transposed weight images refreshed on a side stream: test, same-box A/B
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_loader.py tests/test_gpu_ddp.py -q -m gpu -k "overlapped_transposes or do_train or training_step or early or one_rank" 2>&1 | tail -8 > gpurun_out/r06a/tests.log
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do
  timeout 600 $B > gpurun_out/r06a/ov_$i.json 2> gpurun_out/r06a/ov_$i.err
  timeout 600 $B --no-overlap-transposes > gpurun_out/r06a/sync_$i.json 2> gpurun_out/r06a/sync_$i.err
done
