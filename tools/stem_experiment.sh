#!/bin/bash
# timing-only experiment: what each phase of stem_pool_kernel costs (own instrumented library, never the product one)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stemexp
export VTD_LIB_VARIANT=stemexp
VTD_EXTRA_HIPCC_FLAGS=-DVTD_STEM_EXPERIMENT python video-text-detection-system_amd/build_native.py > gpurun_out/stemexp/build.log 2>&1 || exit 1
for d in ${STEM_VARIANTS:-0 1 2 3 4 5 6 7 8}; do
  VTD_STEM_DEBUG=$d timeout -k 10 200 python bench.py --workload detector --cpu-seconds 0 --sustain-seconds 0 --layers-out gpurun_out/stemexp/layers_dbg$d.json > gpurun_out/stemexp/bench_dbg$d.json 2> gpurun_out/stemexp/err$d.log || true
done
python - <<'PY'
import json
names={0:"product",1:"no pool",2:"no conv tile, no pool",3:"1 of 7 kernel rows",4:"no loads",5:"no stores",6:"setprio",7:"setprio+skew",8:"skew"}
for d in range(9):
    try: t=json.load(open(f'gpurun_out/stemexp/layers_dbg{d}.json'))
    except Exception as e: print(d, e); continue
    for r in t:
        if 'stem' in r['launch'] and r['calls']: print(f"dbg {d} {names[d]:24s} {r['launch'][:50]:50s} {r['ms_total']/r['calls']*1e3:8.1f} us")
PY
