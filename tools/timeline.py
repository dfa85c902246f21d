#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per-kernel durations, per-queue gaps and the busy fraction of the
steady-state window (last 60% of the dispatches)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "dz::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.4):]
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
dur = collections.defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"].split("(")[0][:40]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("window %.1f us, %d dispatches" % ((t1 - t0) / 1e3, len(rows)))
for k, v in dur.items():
    print("  %-42s n=%4d avg %.2f us  sum %.1f us" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3))
# union busy time
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy = 0; cs, ce = ev[0]
ov2 = 0
for s, e in ev[1:]:
    if s <= ce:
        ov2 += min(e, ce) - s
        ce = max(ce, e)
    else:
        busy += ce - cs; cs, ce = s, e
busy += ce - cs
print("busy (union) %.1f us = %.1f%% of window; overlapped %.1f us" % (busy / 1e3, 100.0 * busy / (t1 - t0), ov2 / 1e3))
q = collections.defaultdict(list)
for r in rows:
    q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-30:]))
for k, v in q.items():
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print("queue %s: %d kernels, mean gap %.2f us, median %.2f us" % (k, len(v), sum(gaps) / max(1, len(gaps)) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
if len(sys.argv) > 2:
    for r in rows[:int(sys.argv[2])]:
        print("%8.2f %8.2f q%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"], r["Kernel_Name"].split("(")[0][-40:]))
