cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
( time timeout 1200 python scripts/exp/decomposition_gpu.py 128 gen ) > gpurun_out/r05_decomposition_b128_gen.json 2> gpurun_out/r05_decomposition_b128_gen.err
tail -n 30 gpurun_out/r05_decomposition_b128_gen.json
