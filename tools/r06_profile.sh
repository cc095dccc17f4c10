#!/bin/bash
# Round-6 evidence run on the GPU box: kernel stats + PMC passes over bench.py, summaries into gpurun_out/r06/.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o p -- python $R/bench.py --steps 8 --warmup 4 --no-roofline --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
f=$(find /tmp/p_stats -name "*kernel_stats.csv" | head -1)
cp $f $O/r06_bench_swinL_1024_kernel_stats_all_steps.csv          # rocprofv3's own totals: 12 steps incl. warm-up and graph captures
# steady-state steps only (cut at the optimizer kernel, first 4 steps dropped): what the family summary and bench.py's cross-check read
n=$(python $R/tools/trace_stats.py $(find /tmp/p_stats -name "*kernel_trace.csv" | head -1) 4 $O/r06_bench_swinL_1024_kernel_stats.csv)
echo "{\"kernel_stats_csv\": \"r06_bench_swinL_1024_kernel_stats.csv\", \"steps_in_profile\": $n, \"command\": \"bash tools/r06_profile.sh (rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline; tools/trace_stats.py keeps the $n steady-state steps behind the 4th optimizer launch; r06_bench_swinL_1024_kernel_stats_all_steps.csv = rocprofv3's own --stats over all 12 steps incl. the hipGraph capture passes)\"}" > $O/r06_profile_meta.json
python $R/tools/prof_summary.py $O/r06_bench_swinL_1024_kernel_stats.csv $n > $O/r06_bench_swinL_1024_summary.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -o p -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -o p -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_w.err
ff=$(find /tmp/p_f -name "*counter_collection.csv" | head -1)
fw=$(find /tmp/p_w -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_summary.py $ff $fw 2 2 $O/r06_pmc.json > $O/pmc_summary.txt 2>&1
# bench.py cross-checks its event-clock figures against the committed profile of THIS round: put this run's summaries where it reads them
cp $O/r06_pmc.json $O/r06_profile_meta.json $O/r06_bench_swinL_1024_kernel_stats.csv $R/profiles/
cd $R && python bench.py > $O/r06_bench_line.json 2> $O/bench.err
tail -c 600 $O/r06_bench_line.json
# what BASELINE.md asks to report beside the headline: the shipped configuration's 896^2 and a Swin-T line (same command, other size / backbone)
python bench.py --size 896 --steps 60 --no-cpu-baseline > $O/r06_bench_line_896.json 2>> $O/bench.err
python bench.py --swin T --steps 60 --no-cpu-baseline > $O/r06_bench_line_swinT.json 2>> $O/bench.err
python bench.py --inputs-resident --steps 60 --no-cpu-baseline > $O/r06_bench_line_inputs_resident.json 2>> $O/bench.err
for f in r06_bench_line_896 r06_bench_line_swinT r06_bench_line_inputs_resident; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d['config']['workload'][:60])"; done
# round 6: the same step fed by the product's real loader (generated LVIS-format split; PNG pool and shard store), and the multi-rank path on one GPU
python bench.py --through-loader --steps 60 --warmup 12 --no-cpu-baseline --no-roofline > $O/r06_bench_line_through_loader.json 2>> $O/bench.err
python bench.py --through-loader --loader-shards --steps 60 --warmup 12 --no-cpu-baseline --no-roofline > $O/r06_bench_line_through_loader_shards.json 2>> $O/bench.err
python bench.py --through-loader --loader-shards --loader-scale-range 0.1 2.0 --steps 60 --warmup 20 --no-cpu-baseline --no-roofline > $O/r06_bench_line_through_loader_scale_0p1_2.json 2>> $O/bench.err
python bench.py --gpus 2 --backend gloo --share-device 0 --steps 10 --warmup 4 --no-cpu-baseline --no-roofline > $O/r06_two_rank_gloo_one_gpu.json 2>> $O/bench.err
python bench.py --gpus 2 --backend gloo --share-device 0 --steps 10 --warmup 4 --no-cpu-baseline --no-roofline --allreduce-dtype bf16 > $O/r06_two_rank_gloo_one_gpu_bf16_wire.json 2>> $O/bench.err
for f in r06_bench_line_through_loader r06_bench_line_through_loader_shards r06_bench_line_through_loader_scale_0p1_2 r06_two_rank_gloo_one_gpu r06_two_rank_gloo_one_gpu_bf16_wire; do python -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
# the N = 2 code path fed by the real loader (two loaders, 4 workers each, on the one box)
python bench.py --gpus 2 --backend gloo --share-device 0 --through-loader --loader-shards --workers 4 --steps 10 --warmup 6 --no-cpu-baseline --no-roofline > $O/r06_two_rank_gloo_one_gpu_through_loader.json 2>> $O/bench.err
tail -c 400 $O/r06_two_rank_gloo_one_gpu_through_loader.json
