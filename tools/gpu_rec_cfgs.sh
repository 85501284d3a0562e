#!/bin/bash
# CRNN recogniser alone (272 crops and CROPS2): per-launch table with every conv on one forced tile configuration (VTD_FORCE_CONV_CFG),
# to calibrate the row-count-aware tile choice (vtd_api.cpp: pick_tile_height)
cd $GRAFT_REPO_ROOT
out=gpurun_out/rec_cfgs
mkdir -p $out
export TMPDIR=/tmp
for crops in 272 ${CROPS2:-301}; do
for cfg in table 0 13 12 14 15 1 2; do
  if [ $cfg = table ]; then unset VTD_FORCE_CONV_CFG; else export VTD_FORCE_CONV_CFG=$cfg; fi
  REC_CROPS=$crops timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t_${crops}_$cfg -o run -- python3 tools/lstm_bench.py > $out/log_${crops}_$cfg.txt 2>&1 || { tail -5 $out/log_${crops}_$cfg.txt; exit 1; }
  echo "---- crops $crops cfg $cfg"
  python tools/rec_layers.py $(ls $out/t_${crops}_$cfg/*kernel_trace.csv | head -1) 23 | grep -v "lstm\|compact\|conv1_pool" | cut -c1-120
done
done
