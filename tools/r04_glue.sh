#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
timeout 600 python bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-roofline --copy-sources --inputs-resident > $O/line.json 2> $O/copy_sources.txt
head -45 $O/copy_sources.txt
timeout 900 python bench.py --gpus 2 --backend gloo --share-device 0 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/two_rank.json 2> $O/two_rank.err; tail -c 900 $O/two_rank.json; tail -3 $O/two_rank.err
