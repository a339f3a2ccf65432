#!/bin/bash
# quick single-pass PMC of the association kernel: tools/pmc_quick.sh "COUNTER ..." [bench args]
CNT=${1:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"}
shift
OUT=/tmp/pmcq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CNT --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs "$@" > $OUT/bench.json 2> $OUT/err.log
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmcq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(srl_\w+)', r['Kernel_Name'])
        if m: agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, 'n=', len(next(iter(d.values()))))
PY
