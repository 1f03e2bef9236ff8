"""Summarise a rocprofv3 --kernel-trace CSV: kernels per HW queue / stream, busy time (union of kernel intervals) vs span, and
how many kernels overlap another one -- tells whether trials in flight really run concurrently.

    python scripts/trace_queues.py <dir with *kernel_trace.csv> [tail fraction]
"""
import csv, glob, json, os, sys
from collections import Counter

root = sys.argv[1]
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
path = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(path) as f:
    reader = csv.DictReader(f)
    cols = reader.fieldnames
    for r in reader:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r.get("Stream_Id"), r["Kernel_Name"][:40]))
rows.sort()
t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
cut = t_hi - (t_hi - t_lo) * tail  # steady state: the last part of the run
sel = [r for r in rows if r[0] >= cut]
busy, cur_lo, cur_hi, overlapped = 0, None, None, 0
for s, e, *_ in sel:
    if cur_hi is None or s > cur_hi:
        if cur_hi is not None:
            busy += cur_hi - cur_lo
        cur_lo, cur_hi = s, e
    else:
        overlapped += 1
        cur_hi = max(cur_hi, e)
busy += (cur_hi - cur_lo) if cur_hi is not None else 0
span = sel[-1][1] - sel[0][0]
print(json.dumps(dict(file=os.path.basename(path), columns=cols, kernels=len(rows), steady_kernels=len(sel),
                      span_ms=round(span / 1e6, 2), busy_ms=round(busy / 1e6, 2), sum_kernel_ms=round(sum(e - s for s, e, *_ in sel) / 1e6, 2),
                      overlapped_fraction=round(overlapped / max(len(sel), 1), 3),
                      per_queue=dict(Counter(r[2] for r in sel)), per_stream=dict(Counter(r[3] for r in sel)),
                      mean_gap_us_same_queue=None)))
