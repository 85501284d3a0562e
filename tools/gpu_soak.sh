#!/bin/bash
# repeatability soak (tests/test_gpu_tuning.py), several processes in a row; extra environment assignments as arguments
cd $GRAFT_REPO_ROOT
out=gpurun_out/soak
mkdir -p $out
fails=0
for rep in 1 2 3 4 5 6; do
  env "$@" timeout -k 10 300 python -m pytest tests/test_gpu_tuning.py -x -q -m gpu -k bitwise_repeatable > $out/run_$rep.log 2>&1 || fails=$((fails+1))
done
echo "$fails of 6 failed   $(grep -h 'AssertionError:' $out/run_*.log | cut -c50-120 | tr '\n' ';')"
