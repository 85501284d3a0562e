#!/usr/bin/env python3
"""PMC bytes / launch time for the HBM-bound kernels (profiles/rNN_hbm_bound_kernels.json).

    python tools/hbm_table.py <pmc_traffic_per_launch.json> <kernel_trace.csv> [<layers.json>] <out.json>

bytes  = FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per launch, tools/pmc_summary.py
time   = median launch duration in the rocprofv3 kernel trace of the same bench command (the recogniser and post-process
         streams run beside the detector there), and -- where the launch is a slot of the detector graph -- its HIP-event time with
         the GPU to itself (bench.py --layers-out)
"""
import csv
import json
import statistics
import sys

WANT = {"stem_pool_kernel": "stem_pool", "head_tail_kernel": "head_tail", "pointwise128_kernel": "pointwise128",
        "preprocess_fast_kernel": "preprocess_fast", "conv3x3_c64_persistent_kernel<16, true, false>": "conv3x3_c64_persistent (no residual)",
        "conv3x3_c64_persistent_kernel<16, true, true>": "conv3x3_c64_persistent (+residual)",
        "conv3x3_c64_duo_kernel<16, true, false>": "conv3x3_c64_duo (layer 1, no residual)",
        "conv3x3_c64_duo_kernel<16, true, true>": "conv3x3_c64_duo (layer 1, +residual)", "head_entry_halo256_kernel": "head_entry_halo256",
        "crnn_conv1_pool_kernel": "crnn_conv1_pool", "maxpool_kernel": "maxpool (CRNN)", "lstm_recurrence_kernel": "lstm_recurrence"}
ALONE = {"stem_pool": "stem_pool", "head_tail": "head_tail", "pointwise128": "pointwise128", "head_entry_halo256": "head_entry_halo256"}
ACHIEVABLE_TBS = 6.3   # rocprof-measured streaming ceiling on this part (MI355X_MICROARCH.md), of 8 TB/s nominal


def main():
    traffic = json.load(open(sys.argv[1]))
    layers = json.load(open(sys.argv[3])) if len(sys.argv) > 4 else []
    out_path = sys.argv[-1]
    dur = {}
    with open(sys.argv[2], newline="") as f:
        for r in csv.DictReader(f):
            dur.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    for sym, label in WANT.items():
        t = [(k, v) for k, v in traffic.items() if sym in k]
        d = [v for k, v in dur.items() if sym in k]
        if not t or not d:
            continue
        k, v = max(t, key=lambda kv: kv[1]["launches"])
        mb = v["hbm_read_MB_corrected_x2"] + v["hbm_write_MB"]
        us = statistics.median(max(d, key=len))
        row = {"kernel": label, "hbm_read_MB": v["hbm_read_MB_corrected_x2"], "hbm_write_MB": v["hbm_write_MB"],
               "traced_us_median": round(us, 1), "traced_TBps": round(mb / us, 2), "traced_frac_of_6.3": round(mb / us / ACHIEVABLE_TBS, 2)}
        for lr in layers:
            if lr["calls"] and ALONE.get(label) and lr["launch"].startswith(ALONE[label]):
                a = 1e3 * lr["ms_total"] / lr["calls"]
                row.update(alone_us=round(a, 1), alone_TBps=round(mb / a, 2), **{"alone_frac_of_6.3": round(mb / a / ACHIEVABLE_TBS, 2)})
        rows.append(row)
        print(row)
    json.dump({"achievable_TBps": ACHIEVABLE_TBS, "kernels": rows}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
