#!/bin/bash
# decode of 1088 crops alone: 64 x 64 against 64 x 32 tiles of the decoder's projections on tall live lists (VTD_DEC_GEMM_TALL = row threshold)
cd $GRAFT_REPO_ROOT
for t in ${TALLS:-100000 256 128}; do
  echo "tall >= $t:"; VTD_DEC_GEMM_TALL=$t VTD_TROCR_MAX_CROPS=1280 B=128 REPS=3 timeout -k 10 400 python tools/trocr_stage_bench.py 2>&1 | tail -3
done
