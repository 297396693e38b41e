#!/bin/bash
# usage: timeline.sh OUTDIR  -> per-kernel timeline of one pipelined step (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d $R/$1/trace -- python3 $R/scratch/trace_run.py > $R/$1/trace.log 2>&1
python3 $R/scratch/trace_an.py $R/$1/trace > $R/$1/timeline.txt 2> $R/$1/timeline.err
rm -rf $R/$1/trace
