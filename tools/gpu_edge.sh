#!/bin/bash
# default line: ms per step against the number of timed steps (fill / drain share of the driver's 20-step window)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/edge
for k in ${KS:-10 20 40 80 160}; do
  timeout -k 10 300 python bench.py --steps $k --warmup 5 --cpu-seconds 0 --sustain-seconds 0 --no-profile > gpurun_out/edge/k$k.json 2> gpurun_out/edge/k$k.err || { tail -5 gpurun_out/edge/k$k.err; exit 1; }
  python - <<PY
import json
b=json.load(open("gpurun_out/edge/k$k.json")); print("steps %4d: %.3f ms/step  %.0f frames/s  total %.2f ms" % ($k, b["ms_per_step"], b["value"], b["ms_per_step"]*$k))
PY
done
