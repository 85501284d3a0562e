#!/bin/bash
# Decoder cross-attention on the raw encoder states (VTD_TROCR_XATTN=1) against per-layer key / value projections (0): stage times alone on
# the GPU, the ResNet-18 + Transformer line and configs[4] on one box, per-kernel averages of the decode.
cd $GRAFT_REPO_ROOT
out=gpurun_out/xattn
mkdir -p $out
export TMPDIR=/tmp
for x in 0 1; do
  VTD_TROCR_XATTN=$x timeout -k 10 300 python tools/trocr_stage_bench.py > $out/stages_$x.log 2>&1 || { tail -5 $out/stages_$x.log; exit 1; }
  echo "xattn=$x: $(tail -1 $out/stages_$x.log)"
done
for rep in 1 2; do
for x in 0 1; do
  VTD_TROCR_XATTN=$x timeout -k 10 500 python bench.py --recognizer trocr --steps 16 --warmup 8 --cpu-seconds 0 --sustain-seconds 0 > $out/b_$x$rep.json 2> $out/b_$x$rep.err || { tail -20 $out/b_$x$rep.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/b_$x$rep.json")); r=b["roofline"]; c=r.get("decoder_cross_attention") or {}
print("xattn=$x rep $rep: %.1f frames/s %.1f ms/step | gemm %.0f TFLOP/s | cross-attn %.0f GB/s frac %.3f avg %.1f us at %.0f rows" % (b["value"], b["ms_per_step"], r["achieved"], c.get("achieved") or 0, c.get("frac") or 0, c.get("avg_launch_us") or 0, c.get("avg_live_rows_per_launch") or 0))
PY
done
done
for x in 0 1; do
  VTD_TROCR_XATTN=$x timeout -k 10 900 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 16 --warmup 8 --cpu-seconds 0 --sustain-seconds 0 > $out/c_$x.json 2> $out/c_$x.err || { tail -20 $out/c_$x.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/c_$x.json")); print("configs[4] xattn=$x: %.1f frames/s %.1f ms/step" % (b["value"], b["ms_per_step"]))
PY
done
