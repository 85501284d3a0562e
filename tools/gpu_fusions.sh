#!/bin/bash
# Round-4 visit: parity of the folded downsample projections (conv_igemm DUAL) and the pools fused into the CRNN conv epilogues, then
# A/B of both against the separate launches on one box (detector per-launch tables, default bench line).
cd $GRAFT_REPO_ROOT
out=gpurun_out/fusions
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_detector.py tests/test_gpu_recognizer.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for rep in 1 2; do
for arm in fused separate; do
  if [ $arm = fused ]; then unset VTD_DETECTOR_OPTIONS VTD_RECOGNIZER_OPTIONS; else export VTD_DETECTOR_OPTIONS=fuse_downsample=0 VTD_RECOGNIZER_OPTIONS=fuse_pools=0; fi
  timeout -k 10 200 python bench.py --workload detector --steps 10 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 --layers-out $out/layers_$arm$rep.json > $out/det_$arm$rep.json 2> $out/det_$arm$rep.err || { tail -5 $out/det_$arm$rep.err; exit 1; }
  timeout -k 10 300 python bench.py --cpu-seconds 0 --no-profile > $out/full_$arm$rep.json 2> $out/full_$arm$rep.err || { tail -5 $out/full_$arm$rep.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/det_$arm$rep.json")); f=json.load(open("$out/full_$arm$rep.json"))
rows=[r for r in json.load(open("$out/layers_$arm$rep.json")) if r["calls"]]
print("$arm rep $rep: detector %.0f frames/s | full %.0f frames/s (sustained %s) | detector launches %d, %.0f us summed" % (d["value"], f["value"], f.get("sustained", {}).get("value"), len(rows), sum(1e3*r["ms_total"]/r["calls"] for r in rows)))
PY
done
done
unset VTD_DETECTOR_OPTIONS VTD_RECOGNIZER_OPTIONS
python - <<PY
import json
for arm in ("fused", "separate"):
    print("----", arm)
    for r in json.load(open("$out/layers_%s2.json" % arm)):
        if r["calls"]: print("%8.1f us  %6.0f TFLOP/s  %s" % (1e3*r["ms_total"]/r["calls"], r["tflops"] or 0, r["launch"][:100]))
PY
