#!/bin/bash
# quick A/B, headline + C2, association-only kernel time: tools/ab1.sh lib_a.so lib_b.so ... (libraries under gpurun_in/)
for rep in 1 2; do
for l in "$@"; do
  for wl in "" "--workload C2"; do
    SRL_LIB_PATH=$PWD/gpurun_in/$l python bench.py --steps 40 --no-cpu-baseline --no-configs --no-fused-reduce --no-aux-legs $wl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', '[$wl]', 'assoc-only', round(d['roofline']['avg_launch_ms']*1e3,2), 'us |', round(d['ms_per_esikf_iter']*1e3,1), 'us/iter')"
  done
done
done
