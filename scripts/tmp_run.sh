R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
for dt in bf16 bf16r; do python bench.py --quick --no-roofline --steps 30 --warmup 5 --dtype $dt 2>/dev/null | tail -1 | cut -c1-190; done
python bench.py --mode sample 2>/dev/null | tail -1 | cut -c1-300
