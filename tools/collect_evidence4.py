#!/usr/bin/env python3
"""Copy the summaries of one `tools/gpu_evidence4.sh` visit (gpurun_out/ev4) into profiles/r04_* (the names profiles/README.md lists);
the Transformer workload's PMC entries (dec_*, dense_gemm, trocr_*) of the second visit stay in the one traffic file bench.py looks kernels up in."""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "ev4"), os.path.join(ROOT, "profiles")
COPY = {"bench_full.json": "r04_bench_full.json", "bench_det.json": "r04_bench_detector.json", "bench_full_upload.json": "r04_bench_full_upload.json",
        "bench_full_1080p.json": "r04_bench_full_1080p.json", "bench_full_unfused.json": "r04_bench_full_fusions_off.json",
        "recognizer_launch_table.txt": "r04_recognizer_launch_table.txt",
        "bench_r50_det.json": "r04_bench_r50_detector.json", "layers_det.json": "r04_detector_launch_table.json",
        "layers_r50.json": "r04_r50_detector_launch_table.json", "hbm_bound_kernels.json": "r04_hbm_bound_kernels.json",
        }
for a, b in COPY.items():
    shutil.copyfile(os.path.join(SRC, a), os.path.join(DST, b))
for d, name in (("stats", "r04_full_pipeline_kernel_stats.csv"),):
    shutil.copyfile(glob.glob(os.path.join(SRC, d, "*kernel_stats.csv"))[0], os.path.join(DST, name))
main = json.load(open(os.path.join(SRC, "pmc_traffic_per_launch.json")))
# the Transformer workload's entries (dec_*, dense_gemm, trocr_*) come from the second visit (tools/collect_evidence4b.py): keep what is there
old_path = os.path.join(DST, "r04_pmc_traffic_per_launch.json")
if os.path.exists(old_path):
    for k, v in json.load(open(old_path)).items():
        if any(t in k for t in ("dec_", "dense_gemm", "trocr_")):
            main[k] = v
json.dump(main, open(os.path.join(DST, "r04_pmc_traffic_per_launch.json"), "w"), indent=1, sort_keys=True)
print("copied", len(COPY) + 2, "files")
