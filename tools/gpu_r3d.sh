#!/bin/bash
# round 3: dense-GEMM encoder test, then the Transformer line with / without the dense GEMM and with / without the encode-decode overlap
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3d
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_trocr.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
line() {
python -c "
import json; b=json.load(open('$1')); r=b['roofline']
print('$2: %.1f frames/s  %.1f ms/step  crops/step %.0f | cross-attn %.1f us avg, %.0f GB/s, rows/launch %.1f' % (b['value'], b['ms_per_step'], b['config']['crops_recognized_per_step_rank0'], r['avg_launch_us'], r['achieved'], r['avg_live_rows_per_launch']))"
}
for dense in 0 auto; do for ovl in 1 0; do
  if [ $dense = auto ]; then unset VTD_DENSE_GEMM; else export VTD_DENSE_GEMM=$dense; fi
  VTD_TROCR_DEC_STREAM=$ovl timeout -k 10 400 python bench.py --recognizer trocr --steps 4 --warmup 1 --cpu-seconds 0 --sustain-seconds 0 > $out/b_${dense}_${ovl}.json 2> $out/b_${dense}_${ovl}.err || { tail -20 $out/b_${dense}_${ovl}.err; exit 1; }
  line $out/b_${dense}_${ovl}.json "dense=$dense overlap=$ovl"
done; done
unset VTD_DENSE_GEMM
cd /tmp
VTD_TROCR_DEC_STREAM=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --recognizer trocr --steps 3 --warmup 1 --cpu-seconds 0 --sustain-seconds 0 --no-profile > $GRAFT_REPO_ROOT/$out/prof.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$out/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$out/prof/**/run_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%6.2f%% calls %6s avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("total kernel ms", tot / 1e6)
PY
