#!/bin/bash
# Transformer recogniser: encoder pass of pass k+1 beside the decode of pass k on CU-masked streams.  Parity tests, then the R18 + TrOCR
# line for the back-to-back order and a sweep of the decode's CU share / tickets per pass, then configs[4] at the best setting.
cd $GRAFT_REPO_ROOT
out=gpurun_out/trocr_overlap
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_trocr.py tests/test_gpu_configs.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --recognizer trocr --steps 12 --warmup 4 --cpu-seconds 0 --sustain-seconds 0 > $out/b_$name.json 2> $out/b_$name.err || { tail -20 $out/b_$name.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/b_$name.json")); r=b.get("roofline") or {}
print("$name: %.1f frames/s  %.1f ms/step | %s %.0f %s frac %.3f" % (b["value"], b["ms_per_step"], (r.get("kernel") or "")[:24], r.get("achieved") or 0, r.get("unit"), r.get("frac") or 0))
PY
}
run backtoback VTD_TROCR_OVERLAP=0 || exit 1
run dec64 VTD_TROCR_DEC_CUS=64 || exit 1
run dec80 VTD_TROCR_DEC_CUS=80 || exit 1
run dec96 VTD_TROCR_DEC_CUS=96 || exit 1
run dec112 VTD_TROCR_DEC_CUS=112 || exit 1
run dec96_t3 VTD_TROCR_DEC_CUS=96 VTD_TROCR_PASS_TICKETS=3 VTD_TROCR_MAX_CROPS=1536 || exit 1
run dec128_t4 VTD_TROCR_DEC_CUS=128 VTD_TROCR_PASS_TICKETS=4 VTD_TROCR_MAX_CROPS=2048 || exit 1
