cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
IAN_OPTS=edit_zero_copy=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05g/nozc -o trace -- python scripts/b1_chain_profile.py > gpurun_out/r05g/nozc.log 2>&1
grep -h "seed\|copyBuffer\|dense_bwd\|deconv_out_px" gpurun_out/r05g/nozc/trace_kernel_stats.csv | cut -c1-200
