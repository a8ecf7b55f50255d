#!/bin/bash
# A/B of an environment switch on the headline step:  tools/ab_bench.sh VAR [config]   -> two contract lines (VAR=1 / VAR=0)
VAR=$1; CFG=${2:-c2}
mkdir -p gpurun_out
for v in 1 0; do
  env $VAR=$v PBSED_BENCH_TABLE=1 timeout 300 python bench.py --config $CFG --headline-only --no-cpu-baseline --steps 50 --warmup 10 \
      > gpurun_out/ab_${VAR}_$v.json 2> gpurun_out/ab_${VAR}_$v.err
  python - <<PY
import json
d = json.load(open('gpurun_out/ab_${VAR}_$v.json'))
print('$VAR=$v', d['value'], 'clips/s', d['ms_per_step'], 'ms')
PY
done
