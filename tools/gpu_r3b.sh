#!/bin/bash
# round 3, TrOCR decoder rework: parity tests of the Transformer recogniser + the configs tests, then the three bench lines
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3b
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_trocr.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -x -q -m gpu --durations=6 > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -12 $out/pytest.log
timeout -k 10 400 python bench.py --recognizer trocr --steps 4 --warmup 1 --cpu-seconds 0 --sustain-seconds 0 > $out/bench_r18_trocr.json 2> $out/bench_r18_trocr.err || { tail -20 $out/bench_r18_trocr.err; exit 1; }
python -c "
import json; b=json.load(open('$out/bench_r18_trocr.json')); r=b['roofline']
print('r18+trocr: %.1f frames/s  %.1f ms/step  crops/step %.0f | cross-attn %.1f us avg, %.0f GB/s (%.3f), rows/launch %.1f' % (b['value'], b['ms_per_step'], b['config']['crops_recognized_per_step_rank0'], r['avg_launch_us'], r['achieved'], r['frac'], r['avg_live_rows_per_launch']))"
timeout -k 10 500 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 4 --warmup 1 --cpu-seconds 0 --sustain-seconds 0 > $out/bench_cfg4.json 2> $out/bench_cfg4.err || { tail -20 $out/bench_cfg4.err; exit 1; }
python -c "
import json; b=json.load(open('$out/bench_cfg4.json')); print('cfg4 r50+trocr mixed: %.1f frames/s  %.1f ms/step crops/step %.0f' % (b['value'], b['ms_per_step'], b['config']['crops_recognized_per_step_rank0']))"
timeout -k 10 400 python bench.py > $out/bench_full.json 2> $out/bench_full.err || { tail -20 $out/bench_full.err; exit 1; }
python -c "
import json; b=json.load(open('$out/bench_full.json')); r=b['roofline']; s=b['sustained']; c=b['cpu_baseline']
print('full: %.0f frames/s %.3f ms | in situ %.1f us frac %.3f alone %.3f | sustained %.0f (%.2f) clocks %s -> %s | cpu %s' % (b['value'], b['ms_per_step'], r['avg_launch_us'], r['frac'], r['alone_on_gpu']['frac'], s['value'], s['vs_timed_region'], s['sclk_mhz_before'], s['sclk_mhz_after'], {k: (v['value'] if isinstance(v, dict) else v) for k, v in c.items() if k not in ('sample','unit','kind')}))"
