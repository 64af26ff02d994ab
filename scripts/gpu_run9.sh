cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout 300 python scripts/exp/edit_host_overhead.py 2>&1 | tail -n 2
