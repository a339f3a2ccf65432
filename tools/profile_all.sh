#!/bin/bash
# Run on the GPU box (via gpurun): the un-profiled bench line + rocprofv3 (kernel trace + separate PMC passes) of every
# configuration quoted in DESIGN.md section 5.  Summaries land in gpurun_out/prof_<tag>/ and gpurun_out/bench_final.json.
R=${1:-r06}
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cp gpurun_out/bench_detail.json gpurun_out/bench_final_detail.json
tools/profile_gpu.sh ${R}_headline 10 "" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_headline_unarmed 10 "--no-armed" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_headline_unfused 10 "--no-fused-reduce --no-armed" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_c1 10 "--workload C1" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_c2 10 "--workload C2" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_c3 10 "--workload C3" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_c4 4 "--workload C4" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_init 3 "--frame-id 5" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_headline600 10 "--max-num-residuals 600" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_c2_600 10 "--workload C2 --max-num-residuals 600" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_c3_600 10 "--workload C3 --max-num-residuals 600" > /dev/null 2>&1
tools/profile_gpu.sh ${R}_spread 4 "--workload SPREAD --stream-sweeps 2" > /dev/null 2>&1
for d in gpurun_out/prof_${R}_*; do echo "== $d"; python - "$d" <<'PY'
import json, sys
s = json.load(open(sys.argv[1] + "/summary.json"))
for r in s.get("kernel_stats", [])[:4]:
    print("  ", r["Name"][:40], r.get("Calls"), r.get("AverageNs") or r.get("Average"), r.get("Percentage"))
p = s.get("pmc_per_dispatch", {})
for k, v in p.items():
    print("  pmc", k, {c: round(x) for c, x in list(v.items())[:40] if c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_RD")})
PY
done
