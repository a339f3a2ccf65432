#!/bin/bash
# A/B timing of kernel variants: tools/ab.sh lib_a.so lib_b.so ...   (libraries under gpurun_in/)
for l in "$@"; do
  for rep in 1 2; do
    SRL_LIB_PATH=$PWD/gpurun_in/$l python bench.py --steps 40 --no-cpu-baseline --no-configs --no-fused-reduce --no-aux-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['roofline']['avg_launch_ms']*1e3,2), 'us', round(d['ms_per_esikf_iter']*1e3,1), 'us/iter', round(d['value'],1))"
  done
done
