cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > gpurun_out/r05e/pytest_full.log 2>&1
tail -n 8 gpurun_out/r05e/pytest_full.log
