#!/bin/bash
# SQ counters of dense_gemm_kernel on the encoder's five shapes (two passes of 8 SQ slots; no trace domains beside --pmc):
# how busy are the LDS and the matrix pipe while the kernel runs?
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_dgm
mkdir -p $out
export TMPDIR=/tmp DGM_BENCH_VARIANTS="16"
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$out/p1 -o run -- python3 $R/tools/dense_gemm_bench.py > $R/$out/p1.log 2>&1 || { tail -5 $R/$out/p1.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/$out/p2 -o run -- python3 $R/tools/dense_gemm_bench.py > $R/$out/p2.log 2>&1 || { tail -5 $R/$out/p2.log; exit 1; }
cd $R
python - <<'PY'
import csv, glob, collections, json
res = {}
for d in ("gpurun_out/pmc_dgm/p1", "gpurun_out/pmc_dgm/p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "dense_gemm_kernel" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:80], r["Grid_Size"], r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(" ".join(k), {}).update({c: round(sum(x) / len(x)) for c, x in v.items()})
        res[" ".join(k)]["launches"] = len(next(iter(v.values())))
json.dump(res, open("gpurun_out/pmc_dgm/summary.json", "w"), indent=1)
for k, v in res.items():
    print(k, v)
PY
