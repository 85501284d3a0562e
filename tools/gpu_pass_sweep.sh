#!/bin/bash
# ResNet-18 + Transformer line at B = 32 against the recogniser's pass size (tickets = steps whose crops share one encoder pass + decode)
cd $GRAFT_REPO_ROOT
out=gpurun_out/pass_sweep
mkdir -p $out
for t in ${TS:-4 6 8 12}; do
  VTD_TROCR_PASS_TICKETS=$t VTD_TROCR_MAX_CROPS=$((t * 288)) timeout -k 10 500 python bench.py --recognizer trocr --steps $((3 * t)) --warmup $((2 * t)) --cpu-seconds 0 --sustain-seconds 0 > $out/t$t.json 2> $out/t$t.err || { tail -20 $out/t$t.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/t$t.json")); r=b["roofline"]; c=r.get("decoder_cross_attention") or {}
print("tickets $t: %.1f frames/s %.1f ms/step | gemm %.0f TFLOP/s | cross-attn %.0f GB/s avg %.1f us at %.0f rows" % (b["value"], b["ms_per_step"], r["achieved"], c.get("achieved") or 0, c.get("avg_launch_us") or 0, c.get("avg_live_rows_per_launch") or 0))
PY
done
