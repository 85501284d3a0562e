#!/bin/bash
# HBM-bound kernel work: detector parity tests, then the per-launch table of the detector alone
cd $GRAFT_REPO_ROOT
out=gpurun_out/hbm
mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_detector.py -x -q -m gpu ${PYTEST_K:+-k "$PYTEST_K"} > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 200 python bench.py --workload detector --steps 10 --warmup 2 --cpu-seconds 0 --layers-out $out/layers.json > $out/bench_det.json 2> $out/bench_det.err || { tail -5 $out/bench_det.err; exit 1; }
python - <<PY
import json
b=json.load(open("$out/bench_det.json")); print("detector: %.0f frames/s  %.3f ms/step" % (b["value"], b["ms_per_step"]))
for r in json.load(open("$out/layers.json")):
    if r["calls"]: print("%8.1f us  %s" % (1e3*r["ms_total"]/r["calls"], r["launch"][:100]))
PY
