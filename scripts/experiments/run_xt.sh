mkdir -p gpurun_out/r3d
timeout 300 python scripts/experiments/attic/gemm_pp_check.py > gpurun_out/r3d/check.log 2>&1; tail -2 gpurun_out/r3d/check.log
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or wgrad or rounding" > gpurun_out/r3d/pytest_gemm.log 2>&1; tail -3 gpurun_out/r3d/pytest_gemm.log
for i in 1 2; do
COLD=1 TILE=256 timeout 200 python scripts/gemm_bench.py > gpurun_out/r3d/bench_xt1_$i.log 2>&1
DIC_HIP_LIB=ab/xt0/libdic_hip.so COLD=1 TILE=256 timeout 200 python scripts/gemm_bench.py > gpurun_out/r3d/bench_xt0_$i.log 2>&1
done
paste <(grep TFLOP gpurun_out/r3d/bench_xt1_1.log | awk '{print $1,$2,$3,$4,$(NF-6),$(NF-4)}') <(grep TFLOP gpurun_out/r3d/bench_xt0_1.log | awk '{print $(NF-6),$(NF-4)}') <(grep TFLOP gpurun_out/r3d/bench_xt1_2.log | awk '{print $(NF-6),$(NF-4)}') <(grep TFLOP gpurun_out/r3d/bench_xt0_2.log | awk '{print $(NF-6),$(NF-4)}')
timeout 600 python bench.py --quick > gpurun_out/r3d/bench_quick.log 2>&1; tail -1 gpurun_out/r3d/bench_quick.log | cut -c1-600
DIC_HIP_LIB=ab/xt0/libdic_hip.so timeout 600 python bench.py --quick > gpurun_out/r3d/bench_quick_xt0.log 2>&1; tail -1 gpurun_out/r3d/bench_quick_xt0.log | cut -c1-300
