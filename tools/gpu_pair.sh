#!/bin/bash
# pair-kernel visit: parity of every composed-head-entry configuration, then launch times of 103 vs 105 (survey pass = alone on the GPU, timed = in situ)
cd $GRAFT_REPO_ROOT
out=gpurun_out/pair
mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_detector.py -x -q -m gpu -k "composed_head_entry or fused_fpn" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for cfg in 103 105; do
  VTD_FORCE_CLASSED_CFG=$cfg timeout -k 10 200 python bench.py --cpu-seconds 0 --layers-out $out/layers_$cfg.json > $out/bench_$cfg.json 2> $out/bench_$cfg.err || tail -5 $out/bench_$cfg.err
  python - <<PY
import json
b=json.load(open("$out/bench_$cfg.json")); r=b["roofline"]
print("cfg $cfg: %.0f frames/s | %s | in situ %.1f us frac %.3f | alone %.1f us frac %.3f" % (b["value"], r["kernel"][:24], r["avg_launch_us"], r["frac"], r["alone_on_gpu"]["avg_launch_us"], r["alone_on_gpu"]["frac"]))
PY
done
