#!/bin/bash
# Serial-schedule kernel statistics of one bench configuration under two sets of environment switches (rocprofv3 --kernel-trace --stats):
#   tools/prof_ab.sh <tag> "<bench flags>" "<envA>" "<envB>"     -> gpurun_out/<tag>_{A,B}_kernel_stats.csv
set -o pipefail
R=$GRAFT_REPO_ROOT; tag=$1; flags=$2
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$3" "$4"; do
  n=$([ $i -eq 0 ] && echo A || echo B); i=$((i+1))
  O=$R/gpurun_out/$tag/$n; mkdir -p $O
  env $e AGL_D_STREAMS=0 AGL_G_STREAMS=0 AGL_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/bench.py $flags --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  cp $(find $O -name run_kernel_stats.csv | head -1) $R/gpurun_out/$tag/${n}_kernel_stats.csv
  rm -rf $O
  grep -o '"value": [0-9.]*' $O.log
done
