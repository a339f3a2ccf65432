#!/bin/bash
# final-tree records of the round: GPU suite, the driver's bench command, pipeline probe + kernel statistics, two ranks on one device
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/final
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/final/suite_full.txt 2>&1
grep -E "passed|failed" gpurun_out/final/suite_full.txt | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
cp gpurun_out/bench_detail.json gpurun_out/final/bench_detail.json
timeout 900 python bench.py > gpurun_out/final/bench_default_flags.json 2> gpurun_out/final/bench_default.err
timeout 300 python tools/pipeline_probe.py --frames 6000,24000,65536 --out gpurun_out/final/pipeline_probe.json > gpurun_out/final/pipeline_probe.txt 2>&1
tail -6 gpurun_out/final/pipeline_probe.txt
bash tools/pipeline_kstats.sh 24000 > gpurun_out/final/pipeline_kstats_24k.txt 2>&1
SRL_BENCH_ALL_ON_DEVICE0=1 timeout 600 python bench.py --gpus 2 --transport peer --steps 10 --warmup 3 --no-configs --no-cpu-baseline > gpurun_out/final/bench_two_ranks_peer.json 2> gpurun_out/final/bench_two_ranks.err
python - <<'P'
import json
for n in ("bench", "bench_default_flags", "bench_two_ranks_peer"):
    try:
        d = json.loads(open("gpurun_out/final/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d.get("value"), d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"), (d.get("stream") or {}).get("sweeps_per_s_mean"), [f.get("frames_per_s") for f in (d.get("pipeline") or {}).get("frames", [])], d.get("comm"))
    except Exception as e:
        print(n, "ERR", e)
P
