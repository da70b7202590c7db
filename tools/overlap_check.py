"""How much do the kernels of a multi-stream run overlap?  (rocprofv3 --kernel-trace CSV -> time with 0 / 1 / 2+ kernels in flight)
usage: python3 tools/overlap_check.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void k_", "k_"))]
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s < 20000:   # skip the empty launches
        continue
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, acc = 0, ev[0][0], {}
for tm, d in ev:
    acc[depth] = acc.get(depth, 0) + tm - last
    last = tm
    depth += d
tot = sum(acc.values())
print(f, "queues:", sorted({r["Queue_Id"] for r in rows}))
for k in sorted(acc):
    print("kernels in flight %d: %.1f ms (%.0f %%)" % (k, acc[k] / 1e6, 100.0 * acc[k] / tot))
