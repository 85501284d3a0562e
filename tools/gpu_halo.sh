#!/bin/bash
# per-launch times with the halo-tile kernels forced wherever they apply (1 = first generation, 3 = conv_halo64) against the shipped selection
cd $GRAFT_REPO_ROOT
out=gpurun_out/halo
mkdir -p $out
for c in none 1 3; do
  if [ $c = none ]; then unset VTD_FORCE_HALO; else export VTD_FORCE_HALO=$c; fi
  timeout -k 10 200 python bench.py --workload detector --steps 6 --warmup 2 --cpu-seconds 0 --layers-out $out/layers_$c.json > $out/b_$c.json 2> $out/b_$c.err || { tail -5 $out/b_$c.err; exit 1; }
done
python - <<PY
import json
t={c: json.load(open("$out/layers_%s.json" % c)) for c in ("none","1","3")}
for i,r in enumerate(t["none"]):
    if r["calls"] and ("K=1152" in r["launch"] or "K=2304" in r["launch"] or "K=4608" in r["launch"] or "K=576" in r["launch"]):
        print("%-58s" % r["launch"][:58], " ".join("%7.1f" % (1e3*t[c][i]["ms_total"]/max(t[c][i]["calls"],1)) for c in ("none","1","3")), "|", t["1"][i]["launch"][:22], "|", t["3"][i]["launch"][:22])
PY
