#!/bin/bash
# parity of the detector kernels, then per-launch times alone (survey pass) and the default line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_detector.py tests/test_gpu_pipeline.py -x -q > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python bench.py --workload detector --cpu-seconds 0 --sustain-seconds 3 --layers-out $O/layers.json > $O/bench_detector.json 2> $O/bench_detector.err || { tail $O/bench_detector.err; exit 1; }
timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/bench_full.json 2> $O/bench_full.err || { tail $O/bench_full.err; exit 1; }
python - <<'PY'
import json
O='gpurun_out/r3l'
for r in json.load(open(f'{O}/layers.json')):
    if r['calls']: print(f"{r['launch'][:78]:78s} {r['ms_total']/r['calls']*1e3:8.1f} us {r.get('tflops',0):7.0f} TF")
for n in ('bench_detector','bench_full'):
    d=json.loads(open(f'{O}/{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('alone',''), 'sustained', (d.get('sustained') or {}).get('value'))
PY
