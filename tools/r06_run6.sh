B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --through-loader --loader-shards --workers 8"
$B > gpurun_out/r06_e0.json 2> gpurun_out/r06_e0.err
$B --loader-dev main_threads=4 > gpurun_out/r06_e1.json 2> gpurun_out/r06_e1.err
$B --loader-dev strategy=file_system > gpurun_out/r06_e2.json 2> gpurun_out/r06_e2.err
$B --loader-dev pin=main > gpurun_out/r06_e3.json 2> gpurun_out/r06_e3.err
$B --loader-dev pin=none > gpurun_out/r06_e4.json 2> gpurun_out/r06_e4.err
OMP_NUM_THREADS=4 $B > gpurun_out/r06_e5.json 2> gpurun_out/r06_e5.err
cat /sys/fs/cgroup/cpu.stat > gpurun_out/r06_e_cpustat.txt
python tools/grad_parity_probe.py > gpurun_out/r06_grad_probe.txt 2>&1
