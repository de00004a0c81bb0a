"""Timeline of one steady-state training step from a rocprofv3 kernel trace (usage: python tools/step_timeline.py
<kernel_trace.csv> [step_index]): per queue the busy time and the gaps, and the kernels of the main queue in order with
their start offsets -- where the step's wall time goes when streams overlap."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step boundary = the Morton sort key kernel at the start of every forward
marks = [i for i, r in enumerate(rows) if "sort_key_kernel" in r["Kernel_Name"] or "morton" in r["Kernel_Name"].lower()]
starts = []
for i in marks:
    if not starts or rows[i]["s"] - rows[starts[-1]]["s"] > 5e6:
        starts.append(i)
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 4
a, b = starts[k], starts[k + 1]
step = rows[a:b]
t0 = step[0]["s"]
print(f"step {k}: {len(step)} dispatches, {(rows[b]['s'] - t0) / 1e6:.3f} ms start to start")
queues = {}
for r in step:
    queues.setdefault(r["Queue_Id"], []).append(r)
for q, v in sorted(queues.items(), key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1])):
    busy = sum(r["e"] - r["s"] for r in v) / 1e6
    print(f" queue {q}: {len(v):4d} kernels, busy {busy:7.3f} ms, first +{(v[0]['s'] - t0) / 1e6:.3f}, last end +{(max(r['e'] for r in v) - t0) / 1e6:.3f}")
main = max(queues.values(), key=lambda v: sum(r["e"] - r["s"] for r in v))
print(" main queue, kernels >= 40 us or gaps >= 20 us:")
prev = t0
for r in main:
    gap = (r["s"] - prev) / 1e3
    if gap >= 20:
        print(f"   +{(prev - t0) / 1e6:7.3f}   gap {gap:7.1f} us")
    d = (r["e"] - r["s"]) / 1e3
    if d >= 40:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
        print(f"   +{(r['s'] - t0) / 1e6:7.3f}   {d:7.1f} us  {name}")
    prev = max(prev, r["e"])
