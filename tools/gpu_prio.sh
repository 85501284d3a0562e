#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/prio
mkdir -p $out
for pr in 0 1 0 1; do
  VTD_BENCH_DET_PRIORITY=$pr timeout -k 10 200 python bench.py --cpu-seconds 0 > $out/b$pr.json 2> $out/b$pr.err || { tail -5 $out/b$pr.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/b$pr.json")); r=b["roofline"]
print("prio $pr: %.0f frames/s %.3f ms | in situ %.1f us frac %.3f | alone %.1f us" % (b["value"], b["ms_per_step"], r["avg_launch_us"], r["frac"], r["alone_on_gpu"]["avg_launch_us"]))
PY
done
