#!/bin/bash
# per-kernel stats of the training step (rocprofv3 --kernel-trace --stats), current build and optionally an A/B library: step_kstats.sh [ab-lib-name]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; mkdir -p $O
run() { tag=$1; shift; rm -rf $O/$tag
  env "$@" rocprofv3 --kernel-trace --stats -d $O/$tag --output-format csv -- python $R/bench.py --quick --no-roofline --steps 10 --warmup 3 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); cp $f $O/${tag}_kernel_stats.csv
  python - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6/13:.3f} ms per step (13 steps)")
for r in rows[:16]:
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"\(.*","",n)[:70]
    print(f"{n:70s} calls {int(r['Calls'])//13:4d}/step avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['TotalDurationNs'])/13e6:7.3f} ms/step")
PY
}
run new DIC_WGRAD_STREAM=${WGS:-1}; grep -c "at::native\|rocclr" $O/new_kernel_stats.csv; grep "at::native\|rocclr" $O/new_kernel_stats.csv | cut -c1-120
grep -h '"metric"' $O/new.log | cut -c1-160
if [ -n "$1" ]; then run $1 DIC_WGRAD_STREAM=${WGS:-1} DIC_HIP_LIB=$R/ab/libdic_$1.so; grep -h '"metric"' $O/$1.log | cut -c1-160; fi
