cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
rm -rf gpurun_out/r05_simple gpurun_out/r05_b1 gpurun_out/r05_train gpurun_out/r05_ian
( time timeout 600 python -m pytest tests/test_gpu_reference_pinned.py tests/test_gpu_npe.py -m gpu -x -q -p no:cacheprovider ) > gpurun_out/r05_final_pytest.log 2>&1
tail -n 4 gpurun_out/r05_final_pytest.log
bash scripts/profile_round.sh r05_simple > gpurun_out/r05_simple.log 2>&1
bash scripts/profile_b1.sh r05_b1 > gpurun_out/r05_b1.log 2>&1
bash scripts/profile_train.sh r05_train > gpurun_out/r05_train.log 2>&1
SKIP_FULL_BENCH=1 bash scripts/profile_round.sh r05_ian --arch IAN > gpurun_out/r05_ian.log 2>&1
tail -c 300 gpurun_out/r05_simple/bench.json; echo
