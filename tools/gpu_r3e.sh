#!/bin/bash
# round 3: persistent dense GEMM: microbench (tile orders), TrOCR tests, Transformer bench lines
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3e
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python tools/dense_gemm_bench.py > $out/dgm.log 2>&1 || { tail -20 $out/dgm.log; exit 1; }
tail -10 $out/dgm.log
timeout -k 10 600 python -m pytest tests/test_gpu_trocr.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
line() {
python -c "
import json; b=json.load(open('$1')); r=b['roofline']
print('$2: %.1f frames/s  %.1f ms/step  crops/step %.0f | cross-attn %.1f us avg, %.0f GB/s, rows/launch %.1f' % (b['value'], b['ms_per_step'], b['config']['crops_recognized_per_step_rank0'], r['avg_launch_us'], r['achieved'], r['avg_live_rows_per_launch']))"
}
for ovl in 1 0; do
  VTD_TROCR_DEC_STREAM=$ovl timeout -k 10 400 python bench.py --recognizer trocr --steps 4 --warmup 1 --cpu-seconds 0 --sustain-seconds 0 > $out/b_${ovl}.json 2> $out/b_${ovl}.err || { tail -20 $out/b_${ovl}.err; exit 1; }
  line $out/b_${ovl}.json "r18+trocr overlap=$ovl"
done
