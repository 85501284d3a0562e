#!/bin/bash
# Layer 1 (four 3x3 / 64 -> 64 convolutions, conv3x3_c64_duo_kernel) with and without its HBM traffic, in an instrumented build
# (libvtd_hip_convexp.so, -DVTD_CONV_EXPERIMENT; VTD_C64_DEBUG=1: every pixel block of a workgroup reads and writes one block's
# addresses -- results wrong, time is the point): the ceiling of what fusing a BasicBlock's two convolutions could save.
cd $GRAFT_REPO_ROOT
out=gpurun_out/l1
mkdir -p $out
export VTD_LIB_VARIANT=convexp
VTD_EXTRA_HIPCC_FLAGS=-DVTD_CONV_EXPERIMENT python video-text-detection-system_amd/build_native.py > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
for rep in 1 2; do
for d in 0 1; do
  VTD_C64_DEBUG=$d timeout -k 10 200 python bench.py --workload detector --cpu-seconds 0 --sustain-seconds 0 --steps 10 --warmup 2 --layers-out $out/layers_$d.json > $out/b_$d.json 2> $out/err_$d.log || { tail -5 $out/err_$d.log; exit 1; }
  python - <<PY
import json
rows=[r for r in json.load(open("$out/layers_$d.json")) if r["calls"] and "c64_persistent" in r["launch"]]
print("VTD_C64_DEBUG=$d rep $rep: layer-1 launches", " ".join("%.1f" % (1e3*r["ms_total"]/r["calls"]) for r in rows), "us")
PY
done
done
