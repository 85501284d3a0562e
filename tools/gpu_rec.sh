#!/bin/bash
# the CRNN recogniser alone on the GPU (272 crops): wall time per forward and per-kernel averages
cd $GRAFT_REPO_ROOT
out=gpurun_out/rec
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 200 python tools/lstm_bench.py > $out/wall.log 2>&1 || { tail -5 $out/wall.log; exit 1; }
tail -1 $out/wall.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 tools/lstm_bench.py > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/rec/stats/run_kernel_stats.csv")))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print("%6d calls %8.1f us avg %9.1f us per forward  %s" % (int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/23/1e3, r['Name'][:90]))
PY
python tools/rec_layers.py $(ls $out/stats/*kernel_trace.csv | head -1) 23
