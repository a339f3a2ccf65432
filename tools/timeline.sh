#!/bin/bash
# kernel timeline of a few solves: start/end of every kernel relative to the first, in microseconds
OUT=/tmp/tl; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs "$@" > $OUT/bench.json 2> $OUT/err.log
python - <<'PY'
import csv, glob, re
rows = []
for f in glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if 'srl_assoc' in r['Kernel_Name'] or 'srl_reduce' in r['Kernel_Name']]
sel = sel[-24:]
t0 = int(sel[0]['Start_Timestamp'])
prev_end = None
for r in sel:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    name = re.search(r'srl_\w+', r['Kernel_Name']).group(0)
    gap = '' if prev_end is None else f'gap {(s - prev_end) / 1e3:6.1f}'
    print(f'{name:18s} start {s / 1e3:8.1f} dur {(e - s) / 1e3:6.1f} {gap}')
    prev_end = e
PY
