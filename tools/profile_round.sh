#!/bin/bash
# The round's committed profiles (run on the GPU box: tools/profile_round.sh r03).  rocprofv3 gets the program directly after `--`;
# PMC passes are separate runs with --pmc only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
R=$GRAFT_REPO_ROOT; tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$tag; mkdir -p $O
# the commit the profile describes (the box has no .git): `git rev-parse --short HEAD > .profile_commit` before the gpurun call
export AGL_PROFILE_COMMIT=${AGL_PROFILE_COMMIT:-$(cat $R/.profile_commit 2>/dev/null || echo unknown)}
for cfg in "64_f32x3:" "128_bf16:--res 128 --dtype bf16"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  # (a) the bench command as it runs (chains on several streams): kernel durations include the time a kernel shares the GPU
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$name -o run -- python3 $R/bench.py $fl --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --vary-batch 0 > $O/k_$name.log 2>&1 || exit 1
  cp $O/k_$name/run_kernel_stats.csv $O/bench${name}_kernel_stats.csv
  python3 $R/tools/trace_timeline.py $O/k_$name/run_kernel_trace.csv 6 > $O/bench${name}_timeline.txt 2>&1
  rm -f $O/k_$name/run_kernel_trace.csv
  # (b) the single-stream schedule bench.py's roofline leg times (every kernel alone on the GPU)
  AGL_D_STREAMS=0 AGL_G_STREAMS=0 AGL_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$name -o run -- python3 $R/bench.py $fl --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --vary-batch 0 > $O/s_$name.log 2>&1 || exit 1
  cp $O/s_$name/run_kernel_stats.csv $O/bench${name}_serial_kernel_stats.csv
  grep -o '"value": [0-9.]*' $O/k_$name.log $O/s_$name.log
  # (c) HBM traffic per kernel: FETCH_SIZE / WRITE_SIZE in separate passes + an un-instrumented trace of the same command
  for pass in "f:FETCH_SIZE" "w:WRITE_SIZE"; do
    pt=${pass%%:*}; ctr=${pass#*:}
    AGL_D_STREAMS=0 AGL_G_STREAMS=0 AGL_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/hbm_${pt}_$name -o run -- python3 $R/bench.py $fl --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --vary-batch 0 > $O/hbm_${pt}_$name.log 2>&1 || exit 1
  done
  AGL_D_STREAMS=0 AGL_G_STREAMS=0 AGL_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/hbm_t_$name -o run -- python3 $R/bench.py $fl --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --vary-batch 0 > $O/hbm_t_$name.log 2>&1 || exit 1
  python3 $R/tools/hbm_table.py $O/hbm_f_$name $O/hbm_w_$name $O/hbm_t_$name $O/hbm_kernels_$name.csv $O/hbm_traffic.json ${name} 3 > $O/hbm_table_$name.txt 2>&1
  rm -rf $O/hbm_t_$name $O/hbm_f_$name $O/hbm_w_$name
done
echo profiles done
