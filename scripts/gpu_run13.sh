cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
( time timeout 300 python __graft_entry__.py smoke ) > gpurun_out/r05_smoke.log 2>&1; tail -n 5 gpurun_out/r05_smoke.log
( time IAN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-edit --no-full-ian --train --train-batch 32 ) > gpurun_out/r05_bench_gpus2_gloo.json 2> gpurun_out/r05_bench_gpus2_gloo.err
tail -c 1500 gpurun_out/r05_bench_gpus2_gloo.json; tail -n 5 gpurun_out/r05_bench_gpus2_gloo.err
