#!/bin/bash
# ResNet-50 detector alone: per-launch table with the downsample projections folded into the blocks (default) and as separate launches,
# then the CRNN recogniser alone per launch (row-count-aware tile choice) and the default line.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r50
mkdir -p $out
export TMPDIR=/tmp
for arm in fused separate; do
  if [ $arm = fused ]; then unset VTD_DETECTOR_OPTIONS; else export VTD_DETECTOR_OPTIONS=fuse_downsample=0; fi
  timeout -k 10 300 python bench.py --backbone resnet50 --workload detector --cpu-seconds 0 --sustain-seconds 0 --layers-out $out/layers_$arm.json > $out/b_$arm.json 2> $out/b_$arm.err || { tail -5 $out/b_$arm.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/b_$arm.json")); rows=[r for r in json.load(open("$out/layers_$arm.json")) if r["calls"]]
print("$arm: %.0f frames/s %.3f ms/step | %d launches %.0f us summed | all_mfma %.0f TFLOP/s" % (b["value"], b["ms_per_step"], len(rows), sum(1e3*r["ms_total"]/r["calls"] for r in rows), b["roofline"]["all_mfma_launches_tflops_executed"]))
PY
done
unset VTD_DETECTOR_OPTIONS
python - <<PY
import json
for r in json.load(open("$out/layers_fused.json")):
    if r["calls"]: print("%8.1f us  %6.0f TFLOP/s  %s" % (1e3*r["ms_total"]/r["calls"], r["tflops"] or 0, r["launch"][:100]))
PY
for crops in 272 301; do
REC_CROPS=$crops timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/rec$crops -o run -- python3 tools/lstm_bench.py > $out/rec$crops.log 2>&1 || { tail -5 $out/rec$crops.log; exit 1; }
echo "---- recogniser alone, $crops crops"
python tools/rec_layers.py $(ls $out/rec$crops/*kernel_trace.csv | head -1) 23 | cut -c1-120
done
timeout -k 10 300 python bench.py --cpu-seconds 0 > $out/bench_full.json 2> $out/bench_full.err || { tail -5 $out/bench_full.err; exit 1; }
python - <<PY
import json
b=json.load(open("$out/bench_full.json")); print("full: %.0f frames/s, sustained %.0f" % (b["value"], b["sustained"]["value"]))
PY
