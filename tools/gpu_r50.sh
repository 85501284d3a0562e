#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r50
mkdir -p $out
timeout -k 10 300 python bench.py --backbone resnet50 --workload detector --steps 5 --warmup 2 --cpu-seconds 0 --layers-out $out/layers.json > $out/bench_det.json 2> $out/bench_det.err || { tail -5 $out/bench_det.err; exit 1; }
python - <<PY
import json
b=json.load(open("$out/bench_det.json")); print("r50 detector: %.0f frames/s  %.3f ms/step" % (b["value"], b["ms_per_step"]))
tot=0
for r in json.load(open("$out/layers.json")):
    if r["calls"]:
        us=1e3*r["ms_total"]/r["calls"]; tot+=us
        print("%8.1f us %6.0f TF  %s" % (us, (r["tflops"] or 0), r["launch"][:90]))
print("sum", tot)
PY
