#!/bin/bash
# kernel statistics of the frame pipeline probe (24k-point frames)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $R/tools/pipeline_probe.py --frames ${1:-24000} > /tmp/probe.log 2>&1
tail -3 /tmp/probe.log
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
python $R/tools/kstats.py /tmp/prof | head -40
mkdir -p $R/gpurun_out && cp "$f" $R/gpurun_out/pipeline_kernel_stats.csv
