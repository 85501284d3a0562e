#!/usr/bin/env python3
"""Per-launch table of the CRNN recogniser alone on the GPU from a rocprofv3 --kernel-trace of tools/lstm_bench.py: the dispatches of one
forward in order, each averaged over the forwards of the run (usage: rec_layers.py <kernel_trace.csv> <forwards>)."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
fwd = int(sys.argv[2])
rows = [r for r in rows if "elementwise" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]   # torch's own fills / copies
per = len(rows) // fwd
rows = rows[len(rows) - per * fwd:]
tot = 0.0
for i in range(per):
    d = [int(rows[k * per + i]["End_Timestamp"]) - int(rows[k * per + i]["Start_Timestamp"]) for k in range(fwd)]
    names = {rows[k * per + i]["Kernel_Name"] for k in range(fwd)}
    r = rows[i]
    us = sum(d) / len(d) / 1e3
    tot += us
    print("%2d %8.1f us  grid %-8s wg %-5s %s%s" % (i, us, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")),
                                                   r["Kernel_Name"][:100], "  (!mixed)" if len(names) > 1 else ""))
print("sum %.1f us over %d launches" % (tot, per))
