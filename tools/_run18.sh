cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03i
python tools/glue_by_line.py --top 120 > gpurun_out/r03i/glue_by_line_nograph.txt 2>&1
grep -v Warning gpurun_out/r03i/glue_by_line_nograph.txt | head -140 | cut -c1-200
