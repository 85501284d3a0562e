#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into per-launch HBM traffic (profiles/rNN_pmc_traffic_per_launch.json).

    python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KB summed over the TCC channels; per MI355X_MICROARCH.md's HBM section the
gfx950 FETCH_SIZE under-counts 128-B requests and is corrected x2 (WRITE_SIZE is taken as is).  The median over a
kernel's launches is used so warm-up / autotune launches with other shapes do not skew the figure; kernels that are
launched with several shapes (the convolution instantiations) are additionally split by grid size.
"""
import csv
import glob
import json
import statistics
import sys


def collect(d, counter):
    rows = {}
    for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                rows.setdefault((r["Kernel_Name"], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
    return rows


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for key in sorted(set(fetch) | set(write)):
        name, grid = key
        f = fetch.get(key, [0.0])
        w = write.get(key, [0.0])
        fk, wk = statistics.median(f), statistics.median(w)
        out[f"{name} grid={grid}"] = {
            "launches": max(len(f), len(w)),
            "FETCH_SIZE_KB_median": fk, "WRITE_SIZE_KB_median": wk,
            "hbm_read_MB_corrected_x2": round(fk * 2 * 1024 / 1e6, 2),
            "hbm_write_MB": round(wk * 1024 / 1e6, 2),
        }
    with open(sys.argv[3], "w") as fo:
        json.dump(out, fo, indent=1)
    print("kernels:", len(out))


if __name__ == "__main__":
    main()
