#!/bin/bash
# final-tree records of the round: GPU suite, the driver's bench command (with the committed profiles of THIS kernel: traffic / issue in the
# line), the default-flag bench, pipeline kernel statistics (24k and 256k frames), two ranks on one device with either transport timed
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/suite_full.txt 2>&1
grep -E "passed|failed" gpurun_out/final/suite_full.txt | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
cp gpurun_out/bench_detail.json gpurun_out/final/bench_detail.json
timeout 900 python bench.py > gpurun_out/final/bench_default_flags.json 2> gpurun_out/final/bench_default.err
bash tools/pipeline_kstats.sh 24000 > gpurun_out/final/pipeline_kstats_24k.txt 2>&1
bash tools/pipeline_kstats.sh 262144 > gpurun_out/final/pipeline_kstats_256k.txt 2>&1
make -C tests/fake_rccl > /dev/null 2>&1
for tr in peer rccl; do
  SRL_BENCH_ALL_ON_DEVICE0=1 SRL_BENCH_RCCL_LIBRARY=$R/tests/fake_rccl/libfake_rccl.so timeout 900 python bench.py --gpus 2 --transport $tr --steps 10 --warmup 3 --sharded-config C2 --no-cpu-baseline > gpurun_out/final/bench_two_ranks_$tr.json 2> gpurun_out/final/bench_two_ranks_$tr.err
done
python - <<'P'
import json
for n in ("bench", "bench_default_flags", "bench_two_ranks_peer", "bench_two_ranks_rccl"):
    try:
        d = json.loads(open("gpurun_out/final/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d.get("value"), d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"), (d.get("stream") or {}).get("sweeps_per_s_mean"), [f.get("frames_per_s") for f in (d.get("pipeline") or {}).get("frames", [])], d.get("comm"), d.get("sharded_config"))
    except Exception as e:
        print(n, "ERR", e)
P
