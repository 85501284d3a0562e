#!/bin/bash
# Second look at the encoder pass of pass k+1 beside the decode of pass k on CU-masked streams (VTD_TROCR_OVERLAP=1), now that the decode's
# cross-attention reads a third of the bytes: ResNet-18 + Transformer line per (tickets per pass, CUs of the decode).
cd $GRAFT_REPO_ROOT
out=gpurun_out/overlap2
mkdir -p $out
run() {  # name, steps, warmup, env...
  name=$1; st=$2; wu=$3; shift 3
  env "$@" timeout -k 10 500 python bench.py --recognizer trocr --steps $st --warmup $wu --cpu-seconds 0 --sustain-seconds 0 --no-profile > $out/b_$name.json 2> $out/b_$name.err || { tail -20 $out/b_$name.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/b_$name.json"))
print("$name: %.1f frames/s  %.1f ms/step" % (b["value"], b["ms_per_step"]))
PY
}
run t8_back_to_back 32 16 VTD_TROCR_PASS_TICKETS=8 || exit 1
for cus in ${CUS:-32 48 64 96}; do
  run t8_dec$cus 48 32 VTD_TROCR_OVERLAP=1 VTD_TROCR_DEC_CUS=$cus VTD_TROCR_PASS_TICKETS=8 || exit 1
done
for cus in ${CUS4:-48 64}; do
  run t4_dec$cus 32 16 VTD_TROCR_OVERLAP=1 VTD_TROCR_DEC_CUS=$cus VTD_TROCR_PASS_TICKETS=4 || exit 1
done
