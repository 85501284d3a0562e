#!/bin/bash
# implicit-GEMM tile heights: per-launch times of the detector (alone on the GPU) and the recogniser's forward with configuration
# 12 (208 x 128) / 13 (272 x 128) forced wherever valid, against the shipped selection -- one box
cd $GRAFT_REPO_ROOT
out=gpurun_out/tiles
mkdir -p $out
for c in none 15 13 0; do
  if [ $c = none ]; then unset VTD_FORCE_CONV_CFG; else export VTD_FORCE_CONV_CFG=$c; fi
  timeout -k 10 200 python bench.py --workload detector --steps 6 --warmup 2 --cpu-seconds 0 --layers-out $out/layers_$c.json > $out/b_$c.json 2> $out/b_$c.err || { tail -5 $out/b_$c.err; exit 1; }
  timeout -k 10 200 python tools/lstm_bench.py > $out/rec_$c.log 2>&1 || { tail -5 $out/rec_$c.log; exit 1; }
  echo "cfg $c: recogniser $(tail -1 $out/rec_$c.log)"
done
python - <<PY
import json
t={c: json.load(open("$out/layers_%s.json" % c)) for c in ("none","15","13","0")}
for i,r in enumerate(t["none"]):
    if r["calls"] and "igemm" in r["launch"]:
        print("%-58s" % r["launch"][:58], " ".join("%7.1f" % (1e3*t[c][i]["ms_total"]/max(t[c][i]["calls"],1)) for c in ("none","15","13","0")), "|", t["15"][i]["launch"][11:25], t["13"][i]["launch"][11:21])
PY
