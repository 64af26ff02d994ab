cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
bash scripts/profile_round.sh r05_simple > gpurun_out/r05_simple.log 2>&1
bash scripts/profile_b1.sh r05_b1 > gpurun_out/r05_b1.log 2>&1
bash scripts/profile_train.sh r05_train > gpurun_out/r05_train.log 2>&1
SKIP_FULL_BENCH=1 bash scripts/profile_round.sh r05_ian --arch IAN > gpurun_out/r05_ian.log 2>&1
tail -c 600 gpurun_out/r05_simple/bench.json; echo
tail -n 3 gpurun_out/r05_b1.log gpurun_out/r05_train.log gpurun_out/r05_ian.log
du -sh gpurun_out/r05_simple gpurun_out/r05_b1 gpurun_out/r05_train gpurun_out/r05_ian
