R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
