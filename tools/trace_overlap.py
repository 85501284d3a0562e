#!/usr/bin/env python3
"""Analyse a rocprofv3 kernel trace: per-queue busy time, union busy time, overlap, for the last third of the run
(or the last <window_us> microseconds; a third argument dumps that many kernels)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"]); t_beg = int(rows[0]["Start_Timestamp"])
cut = t_end - int(float(sys.argv[2]) * 1e3) if len(sys.argv) > 2 else t_end - (t_end - t_beg) // 3
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
span = (t_end - int(rows[0]["Start_Timestamp"])) / 1e3
byq = {}
iv = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    byq[r["Queue_Id"]] = byq.get(r["Queue_Id"], 0) + (e - s)
    iv.append((s, e))
iv.sort()
union = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        union += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print(f"window {span:.0f} us; union busy {union / 1e3:.0f} us ({100 * union / 1e3 / span:.1f}%); sum of kernel time {sum(byq.values()) / 1e3:.0f} us")
for q, v in byq.items():
    print(f"  queue {q}: {v / 1e3:.0f} us")
if len(sys.argv) > 3:
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows[: int(sys.argv[3])]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{r['Queue_Id']} {r['Kernel_Name'][:70]}")
