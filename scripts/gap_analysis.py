#!/usr/bin/env python3
"""Read a rocprofv3 kernel-trace CSV and report, for a window of steady-state steps, how much of the wall time has NO kernel
running (launch gaps), how much has exactly one, and how much has two or more (two-stream overlap)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ev.sort()
# steady-state window: from one step's first kernel (the embedding gather) to the next step's = exactly one step
marks = [e for e in ev if "embed_gather" in e[2]]
t0, t1 = marks[-2][0], marks[-1][0]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
pts = []
for s, e, _, _ in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth, last, acc = 0, t0, collections.Counter()
for t, d in pts:
    acc[min(depth, 2)] += t - last
    last = t; depth += d
acc[0] += t1 - last
span = t1 - t0
print(f"step span {span/1e6:.3f} ms, kernels {len(win)}")
for k, name in ((0, "idle (no kernel)"), (1, "one kernel"), (2, ">=2 kernels")):
    print(f"  {name:18s} {acc[k]/1e6:7.3f} ms  {100*acc[k]/span:5.1f}%")
byq = collections.defaultdict(list)
for e in win: byq[e[3]].append(e)
for q, es in byq.items():
    busy = sum(e[1] - e[0] for e in es)
    gaps = [es[i + 1][0] - es[i][1] for i in range(len(es) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"  queue {q}: {len(es)} kernels, busy {busy/1e6:.3f} ms, sum of positive gaps {sum(pos)/1e6:.3f} ms, median gap {sorted(pos)[len(pos)//2]/1e3 if pos else 0:.1f} us")
# biggest gaps on the busiest queue
q = max(byq, key=lambda k: len(byq[k])); es = byq[q]
g = sorted(((es[i + 1][0] - es[i][1], es[i][2][:50], es[i + 1][2][:50]) for i in range(len(es) - 1)), reverse=True)[:12]
for d, a, b in g: print(f"    gap {d/1e3:7.1f} us after {a} -> {b}")
if len(sys.argv) > 2 and sys.argv[2] == "tail":
    print("--- last kernels of the step (offset from step end, us)")
    for s_, e_, name, q_ in sorted(win, key=lambda e: e[1])[-40:]:
        print(f"  q{q_} start {(s_ - t1)/1e3:9.1f} end {(e_ - t1)/1e3:9.1f} dur {(e_-s_)/1e3:7.1f}  {name[:90]}")
if len(sys.argv) > 2 and sys.argv[2] == "head":
    print("--- first kernels of the step (offset from step start, us)")
    for s_, e_, name, q_ in sorted(win)[:45]:
        print(f"  q{q_} start {(s_ - t0)/1e3:9.1f} end {(e_ - t0)/1e3:9.1f} dur {(e_-s_)/1e3:7.1f}  {name[:100]}")
if len(sys.argv) > 2 and sys.argv[2] == "mid":
    print("--- kernels around the middle of the backward (offset from step start, us; both queues, by start time)")
    lnb = [e for e in sorted(win) if "ln_bwd_kernel" in e[2]]
    c = lnb[len(lnb) // 2][0]
    near = [e for e in sorted(win) if abs(e[0] - c) < 700e3]
    for s_, e_, name, q_ in near:
        import re
        nm = re.sub(r"\(anonymous namespace\)::", "", name)
        print(f"  q{q_} start {(s_ - t0)/1e3:9.1f} end {(e_ - t0)/1e3:9.1f} dur {(e_-s_)/1e3:7.1f}  {nm[:80]}")
