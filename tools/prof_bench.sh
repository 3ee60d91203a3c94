#!/bin/bash
# rocprofv3 kernel trace of the bench command (program directly after `--`): tools/prof_bench.sh <tag> <bench args...>
tag=$1; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o run -- python3 bench.py "$@" --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/prof_$tag.log 2>&1
find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_${tag}_kernel_stats.csv
