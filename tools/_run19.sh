cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-330
