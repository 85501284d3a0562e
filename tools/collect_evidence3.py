#!/usr/bin/env python3
"""Copy the summaries of one `tools/gpu_evidence3.sh` visit (gpurun_out/ev3) into profiles/r03_* (the names profiles/README.md lists);
the Transformer workload's PMC entries (dec_*, dense_gemm, trocr_*) are merged into the one traffic file bench.py looks kernels up in."""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "ev3"), os.path.join(ROOT, "profiles")
COPY = {"bench_full.json": "r03_bench_full.json", "bench_det.json": "r03_bench_detector.json", "bench_full_upload.json": "r03_bench_full_upload.json",
        "bench_r18_trocr_b32.json": "r03_bench_r18_trocr_b32.json", "bench_r18_trocr_b64.json": "r03_bench_r18_trocr_b64.json",
        "bench_r18_trocr_b32_unmerged.json": "r03_bench_r18_trocr_b32_unmerged.json",
        "bench_cfg4_b32.json": "r03_bench_cfg4_r50_trocr_mixed_b32.json", "bench_cfg4_b64.json": "r03_bench_cfg4_r50_trocr_mixed_b64.json",
        "bench_r50_det.json": "r03_bench_r50_detector.json", "layers_det.json": "r03_detector_launch_table.json",
        "layers_r50.json": "r03_r50_detector_launch_table.json", "hbm_bound_kernels.json": "r03_hbm_bound_kernels.json",
        "trocr_stages.log": "r03_trocr_stage_times.txt"}
for a, b in COPY.items():
    shutil.copyfile(os.path.join(SRC, a), os.path.join(DST, b))
for d, name in (("stats", "r03_full_pipeline_kernel_stats.csv"), ("stats_trocr", "r03_trocr_pipeline_kernel_stats.csv")):
    shutil.copyfile(glob.glob(os.path.join(SRC, d, "*kernel_stats.csv"))[0], os.path.join(DST, name))
main = json.load(open(os.path.join(SRC, "pmc_traffic_per_launch.json")))
extra = json.load(open(os.path.join(SRC, "pmc_traffic_per_launch_trocr.json")))
for k, v in extra.items():
    if any(t in k for t in ("dec_", "dense_gemm", "trocr_")):
        main[k] = v
json.dump(main, open(os.path.join(DST, "r03_pmc_traffic_per_launch.json"), "w"), indent=1, sort_keys=True)
print("copied", len(COPY) + 3, "files")
