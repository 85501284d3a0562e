#!/bin/bash
# SQ counters of the detector's kernels (one pass, 8 SQ slots; no trace domains beside --pmc)
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_sq
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/p1 -o run -- python3 bench.py --workload detector --cpu-seconds 0 --no-profile --steps 3 --warmup 1 > $out/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $out/p2 -o run -- python3 bench.py --workload detector --cpu-seconds 0 --no-profile --steps 3 --warmup 1 > $out/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_sq/p1","gpurun_out/pmc_sq/p2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d+"/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if any(x in k for x in ("head_entry","persistent","conv_igemm_kernel<256, 128","conv_igemm_kernel<128, 64, 2, 2, 2","stem_pool","head_tail","preprocess")):
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, "launches", len(next(iter(v.values()))))
PY
