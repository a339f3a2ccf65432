#!/bin/bash
# Run on the GPU box (via gpurun).  Collects, for the bench.py command:
#   1. rocprofv3 --kernel-trace --stats            -> per-kernel durations
#   2. separate --pmc passes (no trace domains)     -> SQ / TCC counters of the association kernel
# Summaries land in gpurun_out/prof_$TAG/; copy the ones to be judged into profiles/.
#   tools/profile_gpu.sh TAG [STEPS] ["extra bench.py args"]   e.g.  tools/profile_gpu.sh r02_c2 10 "--workload C2"
TAG=${1:-r01}
STEPS=${2:-10}
SUM=$PWD/gpurun_out/prof_$TAG
OUT=/tmp/prof_raw_$TAG
rm -rf $OUT; mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
EXTRA_ARGS=${3:-}
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-configs --no-aux-legs $EXTRA_ARGS"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_$name.err
done
export SUM
cd $OUT
python - <<'PY'
import csv, glob, collections, os, json, re
out = {}
for f in glob.glob('trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        m = re.search(r'(srl_\w+(<\d+>)?|k_\w+|radix_sort_onesweep|merge_sort_block_merge|__amd_rocclr_\w+)', r['Name'])
        r['Name'] = m.group(1) if m else r['Name'][:80]
    out['kernel_stats'] = rows
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
meta = {}
for f in glob.glob('pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(srl_\w+(<\d+>)?)', r['Kernel_Name'])
        if not m:
            continue
        k = m.group(1)
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[k][r['Counter_Name']].add(r['Dispatch_Id'])
        meta[k] = dict(VGPR=r['VGPR_Count'], AGPR=r['Accum_VGPR_Count'], SGPR=r['SGPR_Count'], LDS=r['LDS_Block_Size'],
                       grid=r['Grid_Size'], wg=r['Workgroup_Size'], scratch=r['Scratch_Size'])
out['pmc_per_dispatch'] = {k: {c: agg[k][c] / max(len(disp[k][c]), 1) for c in agg[k]} for k in agg}
out['pmc_dispatches'] = {k: {c: len(disp[k][c]) for c in agg[k]} for k in agg}
out['kernel_meta'] = meta
import hashlib
h = hashlib.sha256()
root = os.environ.get('GRAFT_REPO_ROOT', '.')
for rel in ("sr_livo_amd/csrc/srl_kernels.hip", "sr_livo_amd/csrc/srl_device.h"):
    h.update(open(os.path.join(root, rel), 'rb').read())
out['kernel_source_sha256'] = h.hexdigest()     # bench.py drops counters whose stamp differs from the tree's (profile_stale)
json.dump(out, open(os.environ.get('SUM', '.') + '/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:7000])
PY
cp $OUT/trace_bench.json $SUM/ 2>/dev/null; find $OUT/trace -name '*stats*.csv' -exec cp {} $SUM/ \;
