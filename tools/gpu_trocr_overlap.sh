#!/bin/bash
# Transformer recogniser: tickets per recogniser pass (back to back) and, for the record, the encoder pass of pass k+1 beside the decode of
# pass k on CU-masked streams (VTD_TROCR_OVERLAP=1).  Parity tests first, then the R18 + TrOCR line per setting on one box.
cd $GRAFT_REPO_ROOT
out=gpurun_out/trocr_overlap
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_trocr.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --recognizer trocr --steps 24 --warmup 8 --cpu-seconds 0 --sustain-seconds 0 > $out/b_$name.json 2> $out/b_$name.err || { tail -20 $out/b_$name.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/b_$name.json")); r=b.get("roofline") or {}
print("$name: %.1f frames/s  %.1f ms/step | %s %.0f %s frac %.3f" % (b["value"], b["ms_per_step"], (r.get("kernel") or "")[:24], r.get("achieved") or 0, r.get("unit"), r.get("frac") or 0))
PY
}


run t4 VTD_TROCR_PASS_TICKETS=4 VTD_TROCR_MAX_CROPS=2048 || exit 1
run t6 VTD_TROCR_PASS_TICKETS=6 VTD_TROCR_MAX_CROPS=3072 || exit 1
run t8 VTD_TROCR_PASS_TICKETS=8 VTD_TROCR_MAX_CROPS=4096 || exit 1

