#!/bin/bash
# Post-process as one workgroup per frame (pp_frame_kernel) against the ten launches it replaces: parity, then the default line on one
# box in alternation (sustained legs are the comparison), then the kernels' own durations in situ.
cd $GRAFT_REPO_ROOT
out=gpurun_out/pp
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_postprocess.py tests/test_gpu_e2e_detector.py tests/test_gpu_pipeline.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for rep in 1 2 3; do
for arm in 1 0; do
  VTD_PP_FUSED=$arm timeout -k 10 300 python bench.py --cpu-seconds 0 > $out/b_$arm$rep.json 2> $out/b_$arm$rep.err || { tail -5 $out/b_$arm$rep.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/b_$arm$rep.json")); r=b["roofline"]
print("VTD_PP_FUSED=$arm rep $rep: %.0f frames/s timed, %.0f sustained | head entry in situ %.1f us frac %.3f" % (b["value"], b["sustained"]["value"], r["avg_launch_us"], r["frac"]))
PY
done
done
cd /tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -o run -- python3 $R/bench.py --cpu-seconds 0 --sustain-seconds 0 --no-profile > $R/$out/stats.log 2>&1 || { tail -5 $R/$out/stats.log; exit 1; }
cd $R
python - <<PY
import csv, glob
rows=list(csv.DictReader(open(glob.glob("$out/stats/*kernel_stats.csv")[0])))
for r in rows:
    if "pp_" in r["Name"]: print("%6d calls %8.1f us avg  %s" % (int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:60]))
PY
rm -rf $out/stats
