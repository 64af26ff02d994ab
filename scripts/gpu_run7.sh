cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_npe.py tests/test_gpu_parity.py tests/test_gpu_reference_pinned.py -m gpu -x -q -p no:cacheprovider -k "not train" ) > gpurun_out/r05f/pytest.log 2>&1
tail -n 4 gpurun_out/r05f/pytest.log
( timeout 300 python scripts/edit_latency.py ) > gpurun_out/r05f/edit.log 2>&1; tail -n 1 gpurun_out/r05f/edit.log
( timeout 300 python scripts/edit_latency.py edit_zero_copy=0 ) > gpurun_out/r05f/edit_nozc.log 2>&1; tail -n 1 gpurun_out/r05f/edit_nozc.log
( timeout 300 python scripts/edit_latency.py ) > gpurun_out/r05f/edit2.log 2>&1; tail -n 1 gpurun_out/r05f/edit2.log
rm -rf gpurun_out/r05_b1; bash scripts/profile_b1.sh r05_b1 > gpurun_out/r05_b1.log 2>&1; tail -n 2 gpurun_out/r05_b1.log
