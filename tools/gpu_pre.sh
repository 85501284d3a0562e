#!/bin/bash
# pre-process kernel: 640 threads per workgroup (product) against 256 (instrumented build libvtd_hip_pre256.so, built here with
# VTD_LIB_VARIANT=pre256 VTD_EXTRA_HIPCC_FLAGS=-DVTD_PRE_NT=256): parity test, kernel time in the detector line, default line
cd $GRAFT_REPO_ROOT
out=gpurun_out/pre
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_detector.py -x -q -m gpu -k "preprocess or transform" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
for v in "" pre256 "" pre256; do
  VTD_LIB_VARIANT=$v timeout -k 10 300 python bench.py --cpu-seconds 0 --sustain-seconds 3 --no-profile > $out/b_$v.json 2> $out/b_$v.err || { tail -5 $out/b_$v.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/b_$v.json")); print("variant '%s': %.0f timed %.0f sustained" % ("$v", b["value"], (b.get("sustained") or {}).get("value") or 0))
PY
done
for v in "" pre256; do
  VTD_LIB_VARIANT=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$v -o run -- python3 bench.py --workload detector --cpu-seconds 0 --sustain-seconds 0 --no-profile > $out/stats_$v.log 2>&1 || { tail -5 $out/stats_$v.log; exit 1; }
  python - <<PY
import csv, glob
rows=list(csv.DictReader(open(glob.glob("$out/stats_$v/**/run_kernel_stats.csv", recursive=True)[0])))
for r in rows:
    if "preprocess" in r["Name"]: print("variant '%s': %s calls %.1f us avg (detector line)" % ("$v", r["Calls"], float(r["AverageNs"])/1e3))
PY
  find $out -name "*kernel_trace.csv" -delete
done
