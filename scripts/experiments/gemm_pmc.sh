#!/bin/bash
# SQ counter passes over scripts/experiments/gemm_pmc.py (8 SQ slots per pass); output under gpurun_out/pmc_gemm/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $O/p1 --output-format csv -- python $R/scripts/experiments/gemm_pmc.py > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_UNALIGNED_STALL -d $O/p2 --output-format csv -- python $R/scripts/experiments/gemm_pmc.py > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES -d $O/p3 --output-format csv -- python $R/scripts/experiments/gemm_pmc.py > $O/p3.log 2>&1
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_gemm"
for p in ("p1", "p2", "p3"):
    fs = glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counter csv"); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        k = (r["Dispatch_Id"], r["Kernel_Name"][:90], r.get("Grid_Size"), r.get("Workgroup_Size"))
        agg.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    with open(f"{O}/{p}_summary.txt", "w") as f:
        for k, v in agg.items():
            if "gemm" not in k[1]: continue
            f.write(f"{k[0]:>5s} {k[1]:90s} grid={k[2]} " + " ".join(f"{n}={x:.4g}" for n, x in v.items()) + "\n")
PY
tail -5 $O/p1.log
