cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( IAN_OPTS=wg_xcd_split=0 timeout 300 python scripts/exp/layer_rates.py 128 ) > gpurun_out/r05c/rates_old.log 2>&1; cp gpurun_out/r05_layer_rates.json gpurun_out/r05c/rates_old.json
( timeout 300 python scripts/exp/layer_rates.py 128 ) > gpurun_out/r05c/rates_new.log 2>&1; cp gpurun_out/r05_layer_rates.json gpurun_out/r05c/rates_new.json
( IAN_OPTS=wg_xcd_split=0 timeout 400 python scripts/exp/train_pair_ms.py ) > gpurun_out/r05c/pair_old.log 2>&1
( timeout 400 python scripts/exp/train_pair_ms.py ) > gpurun_out/r05c/pair_new.log 2>&1
( IAN_OPTS=wg_xcd_split=0 timeout 400 python scripts/exp/train_pair_ms.py ) > gpurun_out/r05c/pair_old2.log 2>&1
( timeout 400 python scripts/exp/train_pair_ms.py ) > gpurun_out/r05c/pair_new2.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py tests/test_gpu_train_step.py -m gpu -x -q -p no:cacheprovider ) > gpurun_out/r05c/pytest.log 2>&1
grep -h "GFLOP\|sums" gpurun_out/r05c/rates_old.log | awk '{print "OLD", $0}' | cut -c1-160
grep -h "GFLOP\|sums" gpurun_out/r05c/rates_new.log | awk '{print "NEW", $0}' | cut -c1-160
tail -1 gpurun_out/r05c/pair_old.log gpurun_out/r05c/pair_new.log gpurun_out/r05c/pair_old2.log gpurun_out/r05c/pair_new2.log
tail -4 gpurun_out/r05c/pytest.log
