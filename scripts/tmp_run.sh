R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "lin_prep or second_bias or centred_layernorm or lo_mean" 2>&1 | tail -3
python scripts/experiments/lin_prep_probe.py 2>&1 | grep -v amdgpu
for i in 1 2; do for dt in bf16 bf16m; do python bench.py --quick --no-roofline --steps 30 --warmup 5 --dtype $dt 2>/dev/null | tail -1 | cut -c1-190; done; done
DIC_CEN=0 python bench.py --quick --no-roofline --steps 30 --warmup 5 --dtype bf16m 2>/dev/null | tail -1 | cut -c1-190
