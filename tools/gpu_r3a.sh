#!/bin/bash
# round 3, first visit: the whole -m gpu suite (new: configs[3]/[4] tests, mixed batches on the device path), sequence lengths of the TrOCR bench workload
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3a
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -15 $out/pytest.log
timeout -k 10 300 python tools/trocr_lengths.py > $out/lengths.log 2>&1 || { tail -20 $out/lengths.log; exit 1; }
tail -4 $out/lengths.log
