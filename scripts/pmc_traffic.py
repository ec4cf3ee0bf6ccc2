#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each its own run with --kernel-trace) into per-kernel HBM traffic.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_in_run> [out.json]
Units/corrections follow MI355X_MICROARCH.md (HBM section): the counters are in KB; on gfx950 FETCH_SIZE under-reports wide
streaming reads by 2x (checked in-run against adamw_kernel, whose traffic is known exactly: 16 B read + 14 B written per
parameter) -- the table prints raw and x2 values and the calibration ratio."""
import collections, csv, json, sys

def load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    return per

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
names = sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] * 2 + write.get(k, [0, 0])[1]))
print(f"{'kernel':64s} {'launches':>8s} {'fetch_MB/launch':>16s} {'fetch_x2':>10s} {'write_MB/launch':>16s}")
tot_f = tot_w = 0.0
gemm = {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0}
folds = {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0}          # split-K slab folds: traffic that exists only because partial sums leave the chip
rows = []
for k in names:
    n = max(fetch.get(k, [0, 0])[0], write.get(k, [0, 0])[0])
    f, w = fetch.get(k, [0, 0.0])[1], write.get(k, [0, 0.0])[1]
    tot_f += f; tot_w += w
    if "gemm_bf16_kernel" in k or "gemm_kernel" in k or "wgrad_group_kernel" in k or "gemm_w4a_kernel" in k:      # = the dic_gemm / dic_wgrad_group calls bench.py counts
        gemm["launches"] += n; gemm["fetch_kb"] += f; gemm["write_kb"] += w
    if "reduce_slabs_kernel" in k or "wgrad_group_fold_kernel" in k:
        folds["launches"] += n; folds["fetch_kb"] += f; folds["write_kb"] += w
    rows.append((k, n, f / n / 1e3, 2 * f / n / 1e3, w / n / 1e3))
for k, n, a, b, c in rows[:40]:
    print(f"{k[:64]:64s} {n:8d} {a:16.1f} {b:10.1f} {c:16.1f}")
print(f"TOTAL per step ({steps} steps): fetch_x2 {2*tot_f/steps/1e6:.2f} GB  write {tot_w/steps/1e6:.2f} GB")
ad = [r for r in rows if "adamw" in r[0]]
if ad:
    k, n, a, b, c = ad[0]
    print(f"calibration: adamw_kernel fetch_x2 {b * n / steps:.1f} MB, write {c * n / steps:.1f} MB per step over {n // steps} launches "
          f"(exact: 16 B read and 14 B written per parameter)")
if len(sys.argv) > 4:
    g = gemm
    out = {"gemm_launches_per_step": g["launches"] / steps, "gemm_fetch_bytes_per_launch_x2": 2e3 * g["fetch_kb"] / g["launches"],
           "gemm_write_bytes_per_launch": 1e3 * g["write_kb"] / g["launches"],
           "gemm_bytes_per_step": (2e3 * g["fetch_kb"] + 1e3 * g["write_kb"]) / steps,
           "fold_launches_per_step": folds["launches"] / steps, "fold_bytes_per_step": (2e3 * folds["fetch_kb"] + 1e3 * folds["write_kb"]) / steps,
           "step_fetch_gb_x2": 2 * tot_f / steps / 1e6, "step_write_gb": tot_w / steps / 1e6,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (each with --kernel-trace); KB units; gfx950 x2 correction on FETCH_SIZE"}
    if len(sys.argv) > 5:
        out["csrc_sha"] = sys.argv[5]          # bench.py quotes this file only for exactly these kernel sources
    json.dump(out, open(sys.argv[4], "w"), indent=1)
