# NMS sweep with the bare resolver loop: tests, step time, decode window
mkdir -p gpurun_out/r06z
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_parity_modules.py tests/test_gpu_loader.py tests/test_gpu_postprocess.py -q -m gpu -k "nms or decode or sampler or training_step or proposal or do_train or inference" 2>&1 | tail -8 > gpurun_out/r06z/tests.log
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline"
for i in 1 2 3; do timeout 600 $B > gpurun_out/r06z/new_$i.json 2> gpurun_out/r06z/new_$i.err; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_n -o p -- python $R/bench.py --steps 6 --warmup 4 --no-roofline --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r06z/prof.err
python $R/tools/decode_window.py $(find /tmp/p_n -name "*kernel_trace.csv" | head -1) 2 > $R/gpurun_out/r06z/window.txt 2>&1
