R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15
