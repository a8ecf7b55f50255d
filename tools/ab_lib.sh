#!/bin/bash
# A/B/... of builds of the library on the headline step (same box, alternating, two rounds):
#   tools/ab_lib.sh "c2 c5" NAME=path.so [NAME=path.so ...]        (path "-" = the in-tree library)
# prints value / ms_per_step per run; contract lines + per-entry-point tables go to gpurun_out/ablib_<name>_<cfg>_<rep>.{json,err}
CFGS=$1; shift
mkdir -p gpurun_out
for cfg in $CFGS; do
  for rep in 1 2; do
    for spec in "$@"; do
      name=${spec%%=*}; lib=${spec#*=}
      if [ "$lib" = "-" ]; then unset PBSED_LIB; else export PBSED_LIB=$(realpath $lib); fi
      out=gpurun_out/ablib_${name}_${cfg}_$rep
      PBSED_BENCH_TABLE=1 timeout 300 python bench.py --config $cfg --headline-only --no-cpu-baseline --steps 50 --warmup 10 > $out.json 2> $out.err || echo "$cfg $name rep$rep: bench failed or timed out (see $out.err)"
      python - <<PY
import json
d = json.load(open('$out.json'))
print('$cfg', '$name', 'rep$rep', d['value'], d['unit'], d['ms_per_step'], 'ms')
PY
    done
  done
done
