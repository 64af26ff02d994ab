cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_train_step.py tests/test_gpu_npe.py -m gpu -x -q -p no:cacheprovider ) > gpurun_out/r05a/pytest_a.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "split_k or tile_config or latent_layer or edit_loop or decoder_forward_cache" ) > gpurun_out/r05a/pytest_b.log 2>&1
( time IAN_DEBUG=1 timeout 600 python scripts/exp/b1_ab.py ) > gpurun_out/r05a/b1_ab.log 2>&1
( time timeout 1500 python scripts/exp/config5_rehearsal.py 8 1024 ) > gpurun_out/r05a/config5.log 2>&1
tail -5 gpurun_out/r05a/pytest_a.log; tail -5 gpurun_out/r05a/pytest_b.log; grep "edit p50" gpurun_out/r05a/b1_ab.log; tail -4 gpurun_out/r05a/config5.log
