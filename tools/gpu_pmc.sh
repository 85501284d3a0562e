#!/bin/bash
# SQ counters of the detector's kernels (one pass, 8 SQ slots; no trace domains beside --pmc)
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_sq; rm -rf $out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/p1 -o run -- python3 bench.py --workload detector --cpu-seconds 0 --no-profile --steps 3 --warmup 1 > $out/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $out/p2 -o run -- python3 bench.py --workload detector --cpu-seconds 0 --no-profile --steps 3 --warmup 1 > $out/p2.log 2>&1
python - <<'PY'
import csv, glob, collections, json
res = {}
for d in ("gpurun_out/pmc_sq/p1", "gpurun_out/pmc_sq/p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if any(x in k for x in ("head_entry", "conv3x3_c64", "conv_halo", "conv_igemm_kernel", "stem_pool", "head_tail", "preprocess", "pointwise", "lstm")):
            res.setdefault(k, {}).update({c: round(sum(x) / len(x)) for c, x in v.items()})
            res[k]["launches"] = len(next(iter(v.values())))
for k, v in res.items():
    # per launch: matrix-pipe busy / LDS-array active as fractions of the kernel's cycles (SQ_BUSY_CYCLES is summed over the 32 shader
    # engines' SQs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs, SQ_LDS_IDX_ACTIVE over the 256 CUs)
    if v.get("SQ_BUSY_CYCLES"):
        cyc = v["SQ_BUSY_CYCLES"] / 32
        v["mfma_pipe_busy_frac"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, 3)
        v["lds_array_active_frac"] = round(v.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc, 3)
        v["lds_bank_conflict_frac_of_lds_cycles"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, v.get("SQ_LDS_IDX_ACTIVE", 0)), 3)
        w = v.get("SQ_WAVE_CYCLES", 0) or 1
        v["wave_cycles_issuing/stalled_at_issue/parked"] = [round(v.get(c, 0) / w, 3) for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")]
    if v.get("SQ_INSTS_MFMA"):
        v["per_mfma_valu/salu/lds/vmem"] = [round(v.get(c, 0) / v["SQ_INSTS_MFMA"], 2) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM")]
    print(k, {c: v[c] for c in v if not c.startswith("SQ_")})
json.dump(res, open("gpurun_out/pmc_sq/summary.json", "w"), indent=1, sort_keys=True)
PY
rm -rf gpurun_out/pmc_sq/p1 gpurun_out/pmc_sq/p2   # (the counter CSVs exceed what gpurun merges back; the summary above is what profiles/ keeps)
