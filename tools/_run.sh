cd $GRAFT_REPO_ROOT
timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_latest.json 2>gpurun_out/bench_latest.err; tail -c 1500 gpurun_out/bench_latest.json
