#!/usr/bin/env python3
"""Copy the summaries of one `tools/gpu_evidence4b.sh` visit (gpurun_out/ev4b) into profiles/r04_*: the Transformer lines after the
cross-attention moved onto the encoder states; that workload's PMC entries replace the older ones in the traffic file bench.py reads."""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "ev4b"), os.path.join(ROOT, "profiles")
COPY = {"bench_r18_trocr_b32.json": "r04_bench_r18_trocr_b32.json", "bench_r18_trocr_b32_t4.json": "r04_bench_r18_trocr_b32_four_tickets_per_pass.json",
        "bench_r18_trocr_b32_kv.json": "r04_bench_r18_trocr_b32_key_value_form.json", "bench_cfg4_b32.json": "r04_bench_cfg4_r50_trocr_mixed_b32.json",
        "trocr_stages.log": "r04_trocr_stage_times.txt", "trocr_stages_kv.log": "r04_trocr_stage_times_key_value_form.txt",
        "lib_gemm.txt": "r04_library_gemm_on_encoder_shapes.txt"}
for a, b in COPY.items():
    shutil.copyfile(os.path.join(SRC, a), os.path.join(DST, b))
shutil.copyfile(glob.glob(os.path.join(SRC, "stats_trocr", "**", "*kernel_stats.csv"), recursive=True)[0], os.path.join(DST, "r04_trocr_pipeline_kernel_stats.csv"))
path = os.path.join(DST, "r04_pmc_traffic_per_launch.json")
main = json.load(open(path))
extra = json.load(open(os.path.join(SRC, "pmc_traffic_per_launch_trocr.json")))
# (the key / value form's cross-attention kernel is not part of this visit's workload and has not changed: its first-visit entries stay)
main = {k: v for k, v in main.items() if "dec_attn_kernel<false" in k or not any(t in k for t in ("dec_", "dense_gemm", "trocr_"))}
for k, v in extra.items():
    if any(t in k for t in ("dec_", "dense_gemm", "trocr_")):
        main[k] = v
json.dump(main, open(path, "w"), indent=1, sort_keys=True)
print("copied", len(COPY) + 2, "files")
