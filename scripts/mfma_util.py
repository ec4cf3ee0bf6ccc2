#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, ...) + its kernel trace.

usage: mfma_util.py <counter_collection.csv> <kernel_trace.csv> <steps_in_run>
MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the SIMDs (16 per v_mfma_f32_16x16x32_bf16,
32 per 32x32x16); GRBM_GUI_ACTIVE counts cycles the GPU was busy under the dispatch.  Utilisation = busy cycles / (active cycles x 1024 SIMDs);
"fraction of the 2.5 PFLOP/s peak" = busy cycles per SIMD per nanosecond of KERNEL DURATION / 2.4 (the peak is quoted at 2.4 GHz): that
column needs no clock estimate.  (Rounds 3-4 printed GRBM_GUI_ACTIVE / duration as a "clock" and built the peak fraction on it; the counter
also runs while the dispatch is being set up, so for short kernels the quotient came out at 3-8 "GHz" -- round-4 review, evidence defect.  The
realised clock inside a launch is measured directly by the clock-probe build, scripts/power_ab.py.)  If the tool reports GRBM_GUI_ACTIVE summed
over the 8 XCDs the script detects that and divides."""
import collections, csv, sys

cnt = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        calls[k] += 1
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Kernel_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
steps = int(sys.argv[3])
tot_busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in cnt.values())
tot_act = sum(v["GRBM_GUI_ACTIVE"] for v in cnt.values())
tot_ns = sum(dur[k] for k in cnt)
scale = 1.0
if tot_ns > 0 and tot_act / tot_ns > 5.0:          # cycles per ns = GHz: > 5 GHz means the counter is a sum over XCDs
    scale = 8.0
N_SIMD = 1024.0
print(f"# MFMA utilisation per kernel, {steps} steps of bench.py --quick (profiled run: clocks ~3 % below an unprofiled one)")
print(f"# util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE{'/8' if scale > 1 else ''} x {int(N_SIMD)} SIMDs); x peak = SQ_VALU_MFMA_BUSY_CYCLES / ({int(N_SIMD)} SIMDs x kernel ns x 2.4 GHz)")
print(f"{'kernel':70s} {'launch/step':>11s} {'ms/step':>8s} {'MFMA util':>9s} {'x peak':>7s}")
rows = sorted(cnt.items(), key=lambda kv: -dur[kv[0]])
for k, v in rows[:24]:
    act = v["GRBM_GUI_ACTIVE"] / scale
    if act <= 0:
        continue
    util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * N_SIMD)
    xpk = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * dur[k] * 2.4) if dur[k] > 0 else 0.0
    name = k.replace("(anonymous namespace)::", "")[:70]
    print(f"{name:70s} {calls[k] / steps:11.1f} {dur[k] / steps / 1e6:8.3f} {util:9.3f} {xpk:7.3f}")
act = tot_act / scale
print(f"{'WHOLE STEP (all kernels, time-weighted)':70s} {'':11s} {tot_ns / steps / 1e6:8.3f} {tot_busy / (act * N_SIMD):9.3f} {tot_busy / (N_SIMD * tot_ns * 2.4):7.3f}")
g_busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for k, v in cnt.items() if "gemm" in k or "wgrad_group_kernel" in k)
g_act = sum(v["GRBM_GUI_ACTIVE"] for k, v in cnt.items() if "gemm" in k or "wgrad_group_kernel" in k) / scale
g_ns = sum(dur[k] for k in cnt if "gemm" in k or "wgrad_group_kernel" in k)
if g_act > 0:
    print(f"{'GEMM kernels only':70s} {'':11s} {g_ns / steps / 1e6:8.3f} {g_busy / (g_act * N_SIMD):9.3f} {g_busy / (N_SIMD * g_ns * 2.4):7.3f}")
