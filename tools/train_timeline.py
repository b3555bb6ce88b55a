#!/usr/bin/env python3
"""Where does a train step's wall time go on the GPU?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --workload train`
and prints, for the last N steps: the span, the busy time (union of kernel intervals), the idle time, the time per
kernel family, and the largest idle gaps with the kernels around them.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload train --steps 6 --warmup 3
    python tools/train_timeline.py DIR 6"""
import csv, glob, os, sys, collections

d, steps = sys.argv[1], int(sys.argv[2])
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
# a step = two launches of the activation-saving forward (coarse, fine): the last `steps` of them
fw = [i for i, r in enumerate(rows) if "mlp_bf16x6_kernel<0, true>" in r[2] or "mlp_f32_kernel<0, true>" in r[2]]
first = fw[-2 * steps]
begin = rows[first][0]
# the step's leading small kernels (audio net, folds) come before its first forward: start at the end of the previous step's last kernel
sel = rows[first:]
span = sel[-1][1] - begin
busy, cur_end, gaps = 0, begin, []
for k, (s, e, n) in enumerate(sel):
    if s > cur_end:
        gaps.append((s - cur_end, sel[k - 1][2] if k else "-", n))
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
fam = collections.Counter()
cnt = collections.Counter()
for s, e, n in sel:
    key = n.split("(")[0][:70]
    fam[key] += e - s
    cnt[key] += 1
print(f"{steps} steps: span {span / 1e6 / steps:.3f} ms/step, busy {busy / 1e6 / steps:.3f}, idle {(span - busy) / 1e6 / steps:.3f} "
      f"({len(gaps) / steps:.0f} gaps/step, {len(sel) / steps:.0f} kernels/step)")
for k, v in fam.most_common(28):
    print(f"  {v / 1e6 / steps:8.3f} ms/step {cnt[k] / steps:6.1f} x  {k}")
gaps.sort(reverse=True)
print("largest gaps (us): ")
for g, a, b in gaps[:14]:
    print(f"  {g / 1e3:8.1f}  after {a.split('(')[0][:50]:50s} before {b.split('(')[0][:50]}")
import statistics
print("gap histogram (us):", {lim: sum(1 for g, _, _ in gaps if g / 1e3 >= lim) for lim in (1, 3, 10, 30, 100)})
