#!/bin/bash
# Whole GPU gate: every -m gpu test in one process, the smoke entry, the default bench line and the detector's per-launch table
cd $GRAFT_REPO_ROOT
out=gpurun_out/full
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -2 $out/smoke.log
timeout -k 10 300 python bench.py > $out/bench_full.json 2> $out/bench_full.err || { tail -5 $out/bench_full.err; exit 1; }
timeout -k 10 200 python bench.py --workload detector --steps 10 --warmup 2 --cpu-seconds 0 --layers-out $out/layers.json > $out/bench_det.json 2> $out/bench_det.err || { tail -5 $out/bench_det.err; exit 1; }
python - <<PY
import json
b=json.load(open("$out/bench_full.json")); r=b["roofline"]
print("full: %.0f frames/s %.3f ms | %s | in situ %.1f us frac %.3f | alone %.1f us frac %.3f" % (b["value"], b["ms_per_step"], r["kernel"][:28], r["avg_launch_us"], r["frac"], r["alone_on_gpu"]["avg_launch_us"], r["alone_on_gpu"]["frac"]))
b=json.load(open("$out/bench_det.json")); print("detector: %.0f frames/s  %.3f ms/step" % (b["value"], b["ms_per_step"]))
for r in json.load(open("$out/layers.json")):
    if r["calls"]: print("%8.1f us  %s" % (1e3*r["ms_total"]/r["calls"], r["launch"][:100]))
PY
