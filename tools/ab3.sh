#!/bin/bash
# A/B of kernel builds over the BASELINE sizes: tools/ab3.sh lib_a.so lib_b.so ...  (libraries under gpurun_in/)
# prints: association kernel alone (reduction in its own kernel) and the fused kernel, us per launch, + us per ESIKF iteration
for rep in 1 2; do
for l in "$@"; do
  for wl in "" "--workload C2" "--workload C3" "--max-num-residuals 600"; do
    A=$(SRL_LIB_PATH=$PWD/gpurun_in/$l python bench.py --steps 40 --no-cpu-baseline --no-configs --no-fused-reduce --no-aux-legs $wl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms']*1e3,2))")
    SRL_LIB_PATH=$PWD/gpurun_in/$l python bench.py --steps 40 --no-cpu-baseline --no-configs --no-aux-legs $wl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', '[$wl]', 'assoc-only', $A, 'us | fused kernel', round(d['roofline']['avg_launch_ms']*1e3,2), 'us |', round(d['ms_per_esikf_iter']*1e3,1), 'us/iter |', round(d['value'],1), 'sweeps/s')"
  done
done
done
