#!/bin/bash
# round 3: TrOCR tests after the attention rewrite, bench lines, and a kernel trace of the Transformer line
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3c
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_trocr.py tests/test_gpu_configs.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for rep in 1 2; do
timeout -k 10 400 python bench.py --recognizer trocr --steps 4 --warmup 1 --cpu-seconds 0 --sustain-seconds 0 > $out/bench_r18_trocr.json 2> $out/bench_r18_trocr.err || { tail -20 $out/bench_r18_trocr.err; exit 1; }
python -c "
import json; b=json.load(open('$out/bench_r18_trocr.json')); r=b['roofline']
print('r18+trocr: %.1f frames/s  %.1f ms/step  crops/step %.0f | cross-attn %.1f us avg, %.0f GB/s (%.3f), rows/launch %.1f' % (b['value'], b['ms_per_step'], b['config']['crops_recognized_per_step_rank0'], r['avg_launch_us'], r['achieved'], r['frac'], r['avg_live_rows_per_launch']))"
done
timeout -k 10 500 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 4 --warmup 1 --cpu-seconds 0 --sustain-seconds 0 > $out/bench_cfg4.json 2> $out/bench_cfg4.err || { tail -20 $out/bench_cfg4.err; exit 1; }
python -c "
import json; b=json.load(open('$out/bench_cfg4.json')); print('cfg4 r50+trocr mixed: %.1f frames/s  %.1f ms/step crops/step %.0f' % (b['value'], b['ms_per_step'], b['config']['crops_recognized_per_step_rank0']))"
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --recognizer trocr --steps 3 --warmup 1 --cpu-seconds 0 --sustain-seconds 0 --no-profile > $GRAFT_REPO_ROOT/$out/prof.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$out/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$out/prof/**/run_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%6.2f%% calls %6s avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("total kernel ms", tot / 1e6)
PY
