#!/bin/bash
# round 3: encoder attention rewrite (V^T pre-pass, 128-query tiles): TrOCR tests, stage timings, kernel averages
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3i
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_trocr.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 300 python tools/trocr_stage_bench.py > $out/stage.log 2>&1 || { tail -20 $out/stage.log; exit 1; }
tail -3 $out/stage.log
cd /tmp
REPS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/trocr_stage_bench.py > $GRAFT_REPO_ROOT/$out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$out/prof/**/run_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print("calls %6s avg %9.1f us  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
