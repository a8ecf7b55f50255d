#!/bin/bash
# A/B of two builds of the library on the headline step (same box, alternating):
#   tools/ab_lib.sh NAME_A=path/a.so NAME_B=path/b.so [config ...]    (path "-" = the in-tree library)
# prints value / ms_per_step per run; the contract lines go to gpurun_out/ablib_<name>_<cfg>_<rep>.json
A=$1; B=$2; shift 2
CFGS=${@:-c2}
mkdir -p gpurun_out
for cfg in $CFGS; do
  for rep in 1 2; do
    for spec in "$A" "$B"; do
      name=${spec%%=*}; lib=${spec#*=}
      if [ "$lib" = "-" ]; then unset PBSED_LIB; else export PBSED_LIB=$(realpath $lib); fi
      out=gpurun_out/ablib_${name}_${cfg}_$rep
      PBSED_BENCH_TABLE=1 python bench.py --config $cfg --headline-only --no-cpu-baseline --steps 50 --warmup 10 > $out.json 2> $out.err
      python - <<PY
import json
d = json.load(open('$out.json'))
print('$cfg', '$name', 'rep$rep', d['value'], d['unit'], d['ms_per_step'], 'ms')
PY
    done
  done
done
