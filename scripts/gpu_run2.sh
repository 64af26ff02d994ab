cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
HL="--steps 50 --warmup 10 --no-cpu-baseline --no-edit --no-train --no-full-ian"
( time timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pinned.py -m gpu -x -q -p no:cacheprovider -k "not train" ) > gpurun_out/r05b/pytest.log 2>&1
export IAN_TUNE_CACHE=$PWD/gpurun_out/r05b/tune.txt
( timeout 300 python bench.py $HL ) > gpurun_out/r05b/bench_default.json 2> gpurun_out/r05b/bench_default.err
( IAN_OPTS=dec_out_wgs=512 timeout 300 python bench.py $HL ) > gpurun_out/r05b/bench_wgs512.json 2> gpurun_out/r05b/bench_wgs512.err
( timeout 300 python bench.py $HL ) > gpurun_out/r05b/bench_default2.json 2> gpurun_out/r05b/bench_default2.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05b/trace -o trace -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-edit --no-train --no-full-ian > gpurun_out/r05b/trace.log 2>&1
IAN_OPTS=dec_out_wgs=512 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05b/trace512 -o trace -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-edit --no-train --no-full-ian > gpurun_out/r05b/trace512.log 2>&1
unset IAN_TUNE_CACHE
( time timeout 400 python scripts/exp/layer_rates.py 128 ) > gpurun_out/r05b/layer_rates.log 2>&1
tail -3 gpurun_out/r05b/pytest.log
for f in default wgs512 default2; do python -c "import json,sys; d=json.loads(open('gpurun_out/r05b/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
for t in trace trace512; do grep -h "conv1_mfma\|deconv_small" gpurun_out/r05b/$t/*kernel_stats.csv; done
cat gpurun_out/r05b/layer_rates.log | grep -v "^$" | tail -16
