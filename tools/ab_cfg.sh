for rep in 1 2; do
for l in lib_overlap.so lib_nooverlap.so; do
  for wl in "--workload C3" "--workload C2" "--max-num-residuals 600" ""; do
    SRL_LIB_PATH=$PWD/gpurun_in/$l python bench.py --steps 40 --no-cpu-baseline --no-configs --no-aux-legs $wl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', '$wl', round(d['roofline']['avg_launch_ms']*1e3,2), 'us', round(d['ms_per_esikf_iter']*1e3,1), 'us/iter', round(d['value'],1))"
  done
done
done
