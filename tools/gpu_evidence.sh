#!/bin/bash
# Evidence visit: default bench (both workloads), sustained run with the shader clock sampled beside it, kernel-trace stats, the two PMC
# traffic passes and the SQ instruction-mix passes of the same command.  Outputs under gpurun_out/ev; the summaries go to profiles/.
cd $GRAFT_REPO_ROOT
out=gpurun_out/ev
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --layers-out $out/layers_full.json > $out/bench_full.json 2> $out/bench_full.err || { tail -5 $out/bench_full.err; exit 1; }
cut -c1-300 $out/bench_full.json
timeout -k 10 300 python bench.py --workload detector --cpu-seconds 0 --layers-out $out/layers_det.json > $out/bench_det.json 2> $out/bench_det.err || { tail -5 $out/bench_det.err; exit 1; }
cut -c1-200 $out/bench_det.json
( for i in $(seq 1 60); do date +%s.%N; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*'; rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -1; sleep 0.5; done ) > $out/clock_samples.txt 2>&1 &
sampler=$!
timeout -k 10 400 python bench.py --steps 4000 --warmup 20 --cpu-seconds 0 --no-profile > $out/bench_sustained.json 2> $out/bench_sustained.err
kill $sampler 2>/dev/null
cut -c1-300 $out/bench_sustained.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --cpu-seconds 0 --no-profile > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 bench.py --cpu-seconds 0 --no-profile --steps 4 --warmup 1 > $out/pmc_fetch.log 2>&1 || { tail -5 $out/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 bench.py --cpu-seconds 0 --no-profile --steps 4 --warmup 1 > $out/pmc_write.log 2>&1 || { tail -5 $out/pmc_write.log; exit 1; }
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic_per_launch.json
python tools/hbm_table.py $out/pmc_traffic_per_launch.json $(ls $out/stats/*kernel_trace.csv | head -1) $out/layers_det.json $out/hbm_bound_kernels.json
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -o run -- python3 bench.py --workload detector --cpu-seconds 0 --no-profile --steps 3 --warmup 1 > $out/pmc_sq.log 2>&1 || { tail -5 $out/pmc_sq.log; exit 1; }
python - <<'PY'
import csv, glob, collections, json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/ev/pmc_sq/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={}
for k,v in acc.items():
    if any(x in k for x in ("head_entry","persistent","conv_igemm","stem_pool","head_tail","preprocess","pointwise")):
        row={c: round(sum(x)/len(x)) for c,x in v.items()}
        row["launches"]=len(next(iter(v.values())))
        if row.get("SQ_INSTS_MFMA"): row["valu_per_mfma"]=round(row["SQ_INSTS_VALU"]/row["SQ_INSTS_MFMA"],2)
        if row.get("SQ_BUSY_CYCLES"): row["mfma_pipe_busy_frac"]=round(row.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/1024/(row["SQ_BUSY_CYCLES"]/32),3)
        out[k]=row
json.dump(out, open("gpurun_out/ev/pmc_sq_summary.json","w"), indent=1)
for k,r in out.items(): print(k[:60], r.get("valu_per_mfma"), r.get("mfma_pipe_busy_frac"))
PY
ls $out/stats | head -5
