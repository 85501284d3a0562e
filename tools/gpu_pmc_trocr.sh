#!/bin/bash
# SQ counters of the Transformer recogniser's kernels over tools/trocr_stage_bench.py (two passes of 8 SQ slots; no trace domains)
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_trocr
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$out/p1 -o run -- python3 $R/tools/trocr_stage_bench.py > $R/$out/p1.log 2>&1 || { tail -5 $R/$out/p1.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $R/$out/p2 -o run -- python3 $R/tools/trocr_stage_bench.py > $R/$out/p2.log 2>&1 || { tail -5 $R/$out/p2.log; exit 1; }
cd $R
python - <<'PY'
import csv, glob, collections, json
res = {}
for d in ("gpurun_out/pmc_trocr/p1", "gpurun_out/pmc_trocr/p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {}).update({c: round(sum(x) / len(x)) for c, x in v.items()})
        res[k]["launches"] = len(next(iter(v.values())))
out = {}
for k, v in res.items():
    o = {"launches": v["launches"]}
    if v.get("SQ_BUSY_CYCLES"):
        cyc = v["SQ_BUSY_CYCLES"] / 32
        o["mfma_pipe_busy_frac"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, 3)
        o["lds_array_active_frac"] = round(v.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc, 3)
        o["lds_bank_conflict_frac_of_lds_cycles"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, v.get("SQ_LDS_IDX_ACTIVE", 0)), 3)
        w = v.get("SQ_WAVE_CYCLES", 0) or 1
        o["wave_cycles_issuing/stalled_at_issue/parked"] = [round(v.get(c, 0) / w, 3) for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")]
    if v.get("SQ_INSTS_MFMA"):
        o["per_mfma_valu/salu/lds/vmem"] = [round(v.get(c, 0) / v["SQ_INSTS_MFMA"], 2) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM")]
    else:
        o["insts_valu/salu/lds/vmem"] = [v.get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM")]
    out[k] = o
    print(k, o)
json.dump(out, open("gpurun_out/pmc_trocr/summary.json", "w"), indent=1, sort_keys=True)
PY
rm -rf $out/p1 $out/p2
rm -rf gpurun_out/pmc_trocr/p1 gpurun_out/pmc_trocr/p2
