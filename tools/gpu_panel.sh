#!/bin/bash
# K-panel-major GEMM inputs (VTD_DENSE_PANEL=1) against row-major (0) on one box, alternating: the ResNet-18 + Transformer line at 12 and 4
# tickets per pass, and the encoder pass alone at 272 / 1088 crops
cd $GRAFT_REPO_ROOT
out=gpurun_out/panel
mkdir -p $out
for rep in 1 2; do
for p in 0 1; do
  VTD_DENSE_PANEL=$p timeout -k 10 600 python bench.py --recognizer trocr --steps 36 --warmup 24 --cpu-seconds 0 --sustain-seconds 0 --no-profile > $out/t12_$p$rep.json 2> $out/t12_$p$rep.err || { tail -20 $out/t12_$p$rep.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/t12_$p$rep.json")); print("12 tickets, panel=$p rep $rep: %.1f frames/s %.1f ms/step" % (b["value"], b["ms_per_step"]))
PY
done
done
for p in 0 1; do
  echo "1088 crops alone, panel=$p: $(VTD_DENSE_PANEL=$p VTD_TROCR_MAX_CROPS=1280 B=128 REPS=2 python tools/trocr_stage_bench.py 2>&1 | tail -1)"
  echo "3264 crops alone, panel=$p: $(VTD_DENSE_PANEL=$p VTD_TROCR_MAX_CROPS=3456 B=384 REPS=1 python tools/trocr_stage_bench.py 2>&1 | tail -1)"
done
