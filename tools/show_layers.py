#!/usr/bin/env python3
"""Print a bench.py --layers-out table: per-launch average microseconds and executed TFLOP/s."""
import json
import sys

d = json.load(open(sys.argv[1]))
tot = 0.0
for r in d:
    c = r["calls"]
    if not c:
        continue
    us = r["ms_total"] / c * 1e3
    tot += us
    print(f"{r['launch'][:86]:86s} {us:8.1f} us {r['gmac_total'] / c:8.2f} GMAC  {round(r['tflops']) if r['tflops'] else '-'}")
print(f"sum {tot:.1f} us")
