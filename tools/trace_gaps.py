"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: start offset, duration and the gap to the
previous kernel end (any stream).   python tools/trace_gaps.py <kernel_trace.csv> [step_index_from_end]"""
import csv
import sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
# a step starts at its first forward chain kernel: the first rec_fwd after a rec_bwd (stacked layers launch several of each)
starts, seen_bwd = [], True
for i, r in enumerate(rows):
    if "rec_bwd" in r[2]:
        seen_bwd = True
    elif "rec_fwd" in r[2] and seen_bwd:
        starts.append(i)
        seen_bwd = False
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
i0, i1 = starts[-k - 1], starts[-k]
t0 = rows[i0][0]
busy_end = t0
print("step span %.1f us" % ((rows[i1][0] - t0) / 1e3))
chains = sum(e - s for s, e, n, q in rows[i0:i1] if "rec_fwd" in n or "rec_bwd" in n)
last_end = max(e for s, e, n, q in rows[i0:i1])
print("chain kernels %.1f us, outside them %.1f us (first kernel start -> last kernel end %.1f us)"
      % (chains / 1e3, (last_end - t0 - chains) / 1e3, (last_end - t0) / 1e3))
for s, e, n, q in rows[i0:i1]:
    print("%8.1f +%7.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - busy_end) / 1e3, q, n[:70]))
    busy_end = max(busy_end, e)
