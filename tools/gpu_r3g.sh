#!/bin/bash
# round 3: decoder tail variants (short row tiles, 16-wave cross-attention, LN parameters hoisted): tests, steady-state bench lines (8 steps)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3g
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python tools/trocr_stage_bench.py > $out/stage.log 2>&1; tail -3 $out/stage.log
timeout -k 10 600 python -m pytest tests/test_gpu_trocr.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
line() {
python -c "
import json; b=json.load(open('$1')); r=b['roofline']
print('$2: %.1f frames/s  %.1f ms/step  crops/step %.0f | cross-attn %.1f us avg, %.0f GB/s, rows/launch %.1f' % (b['value'], b['ms_per_step'], b['config']['crops_recognized_per_step_rank0'], r['avg_launch_us'], r['achieved'], r['avg_live_rows_per_launch']))"
}
timeout -k 10 400 python bench.py --recognizer trocr --steps 8 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/b32.json 2> $out/b32.err || { tail -20 $out/b32.err; exit 1; }
line $out/b32.json "r18+trocr B=32"


VTD_TROCR_MAX_CROPS=1024 timeout -k 10 500 python bench.py --recognizer trocr --batch 64 --steps 6 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/b64.json 2> $out/b64.err || { tail -20 $out/b64.err; exit 1; }
line $out/b64.json "r18+trocr B=64"
timeout -k 10 500 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 8 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/cfg4.json 2> $out/cfg4.err || { tail -20 $out/cfg4.err; exit 1; }
line $out/cfg4.json "cfg4 r50+trocr mixed B=32"
VTD_TROCR_MAX_CROPS=1024 timeout -k 10 600 python bench.py --backbone resnet50 --recognizer trocr --mixed --batch 64 --steps 6 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/cfg4_64.json 2> $out/cfg4_64.err || { tail -20 $out/cfg4_64.err; exit 1; }
line $out/cfg4_64.json "cfg4 r50+trocr mixed B=64"
