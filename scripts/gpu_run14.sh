cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
( time timeout 600 python bench.py ) > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
tail -c 400 gpurun_out/r05_bench_final.json
